"""Synthetic inputs of BASELINE.json's configs 2-5, as SURVEY.md section 8(d) fixes them (seeds, lengths, scoring).

Shared by bench.py (`--config N`) and the full-shape parity tests; numpy only, vectorised so that one GPU's
share of an 8-GPU config (1.25M pairs) is generated in seconds.  `rank` shifts the seed so every rank of a
weak-scaling run holds its own batch.  Everything is returned in the packed layout of include/parasail_amd.h
(uint8 buffer + int64 offsets[n+1]).
"""
import numpy as np

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)

# scoring per config: (alphabet or builtin name, match, mismatch, open, extend)
CFG2 = dict(seed=20260001, n=1_000_000, len=150, matrix=("ACGT", 2, -3), open=5, ext=2, name="sw_striped_16")
CFG3 = dict(seed=20260003, n=100_000, qlen=300, rlo=4500, rhi=5000, matrix="blosum62", open=11, ext=1,
            name="nw_stats_striped_profile_16")
CFG4 = dict(seed=20260004, n=10_000_000, len=250, sub=0.10, indel=0.02, matrix=("ACGT", 2, -3), open=5, ext=2,
            name="sg_trace_striped_16")
CFG5 = dict(seed=20260005, n=10_000_000, qlen=1000, rlo=500, rhi=5000, plant_frac=0.01, sub=0.05, indel=0.01,
            matrix=("ACGT", 2, -3), open=5, ext=2, name="sw_striped_profile_sat")


def uniform_offsets(n, length):
    return np.arange(n + 1, dtype=np.int64) * length


def make_cfg2(n=CFG2["n"], rank=0):
    """1M x (150 x 150) i.i.d. DNA: queries first, then references, 0..3 -> ACGT."""
    rng = np.random.default_rng(CFG2["seed"] + rank)
    L = CFG2["len"]
    q = DNA[rng.integers(0, 4, size=(n, L), dtype=np.uint8)].reshape(-1)
    r = DNA[rng.integers(0, 4, size=(n, L), dtype=np.uint8)].reshape(-1)
    off = uniform_offsets(n, L)
    return q, off, r, off.copy()


def make_cfg3(n=CFG3["n"], rank=0):
    """One 300-aa query (uniform over the 20 residues) and n references of 4 500-5 000 aa.
    Returns (query bytes, rbuf, roff)."""
    q = AA[np.random.default_rng(CFG3["seed"]).integers(0, 20, size=CFG3["qlen"])].tobytes()   # the same query on every rank
    rng = np.random.default_rng(CFG3["seed"] + 1000 + rank)
    lens = rng.integers(CFG3["rlo"], CFG3["rhi"] + 1, size=n)
    roff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=roff[1:])
    rbuf = AA[rng.integers(0, 20, size=int(roff[-1]), dtype=np.uint8)]
    return q, rbuf, roff


def related_fixed(rng, q2d, sub, indel, alphabet=DNA):
    """Related-pair generator, vectorised: every row of q2d [n, L] gets `sub` substitutions and `indel`
    single-symbol indels (half deletions, half insertions), then is truncated / padded with random symbols
    back to L columns ("ref re-padded/truncated")."""
    n, L = q2d.shape
    if n > 65536:                       # chunks keep the int32 index temporaries small
        return np.concatenate([related_fixed(rng, q2d[a:a + 65536], sub, indel, alphabet) for a in range(0, n, 65536)])
    na = len(alphabet)
    r = q2d.copy()
    # (16-bit uniform draws against integer thresholds: several times faster than float64 draws)
    smask = rng.integers(0, 65536, size=(n, L), dtype=np.uint16) < int(sub * 65536)
    r[smask] = alphabet[rng.integers(0, na, size=int(smask.sum()), dtype=np.uint8)]
    u = rng.integers(0, 65536, size=(n, L), dtype=np.uint16)
    t_del, t_ins = int(indel / 2 * 65536), int(indel * 65536)
    counts = np.ones((n, L), dtype=np.int8)
    counts[u < t_del] = 0                                       # deletion
    ins = (u >= t_del) & (u < t_ins)
    counts[ins] = 2                                             # a random symbol inserted in front
    flat = np.repeat(r.reshape(-1), counts.reshape(-1))
    # the first copy of every doubled symbol becomes the inserted random symbol
    ends = np.cumsum(counts.reshape(-1), dtype=np.int32)
    ins_pos = ends[ins.reshape(-1)] - 2
    flat[ins_pos] = alphabet[rng.integers(0, na, size=len(ins_pos), dtype=np.uint8)]
    row_len = counts.sum(axis=1, dtype=np.int32)
    row_start = np.concatenate([[0], np.cumsum(row_len, dtype=np.int32)[:-1]]).astype(np.int32)
    cols = np.arange(L, dtype=np.int32)[None, :]
    valid = cols < row_len[:, None]
    idx = np.minimum(row_start[:, None] + cols, np.int32(len(flat) - 1))
    out = flat[idx]
    pad = ~valid
    out[pad] = alphabet[rng.integers(0, na, size=int(pad.sum()), dtype=np.uint8)]
    return out


def make_cfg4(n=CFG4["n"] // 8, rank=0):
    """n pairs of 250 x 250 related DNA (10 % substitutions, 2 % indels)."""
    rng = np.random.default_rng(CFG4["seed"] + rank)
    L = CFG4["len"]
    q = DNA[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    r = related_fixed(rng, q, CFG4["sub"], CFG4["indel"])
    off = uniform_offsets(n, L)
    return q.reshape(-1), off, r.reshape(-1), off.copy()


def mutate_row(rng, seq, sub, indel, alphabet=DNA):
    """One sequence (uint8 array) -> noisy copy, length free."""
    na = len(alphabet)
    r = seq.copy()
    smask = rng.random(len(r)) < sub
    r[smask] = alphabet[rng.integers(0, na, size=int(smask.sum()))]
    u = rng.random(len(r))
    counts = np.ones(len(r), dtype=np.int64)
    counts[u < indel / 2] = 0
    ins = (u >= indel / 2) & (u < indel)
    counts[ins] = 2
    out = np.repeat(r, counts)
    ends = np.cumsum(counts)
    pos = ends[ins] - 2
    out[pos] = alphabet[rng.integers(0, na, size=len(pos))]
    return out


def make_cfg5(n=CFG5["n"] // 8, rank=0):
    """One 1 kbp query; n references with log-uniform lengths in [500, 5000]; a noisy copy of the query
    (5 % substitutions, 1 % indels) overwrites a random window of 1 % of the references (lengths unchanged,
    a reference shorter than the copy takes its prefix).  Returns (query bytes, rbuf, roff, planted index array)."""
    q = DNA[np.random.default_rng(CFG5["seed"]).integers(0, 4, size=CFG5["qlen"])]
    rng = np.random.default_rng(CFG5["seed"] + 1000 + rank)
    lens = np.exp(rng.uniform(np.log(CFG5["rlo"]), np.log(CFG5["rhi"]), size=n)).astype(np.int64)
    roff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=roff[1:])
    rbuf = DNA[rng.integers(0, 4, size=int(roff[-1]), dtype=np.uint8)]
    planted = np.sort(rng.choice(n, size=max(1, int(n * CFG5["plant_frac"])), replace=False))
    for k in planted:
        copy = mutate_row(rng, q, CFG5["sub"], CFG5["indel"])
        L = int(lens[k])
        m = min(L, len(copy))
        pos = int(rng.integers(0, L - m + 1))
        rbuf[roff[k] + pos: roff[k] + pos + m] = copy[:m]
    return q.tobytes(), rbuf, roff, planted
