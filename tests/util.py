"""Shared helpers for the parity tests (seeded synthetic inputs, score-from-CIGAR recomputation)."""
import re

import numpy as np

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)


def random_seqs(rng, n, lo, hi, alphabet=DNA):
    lens = rng.integers(lo, hi + 1, size=n)
    return [alphabet[rng.integers(0, len(alphabet), size=int(l))].tobytes() for l in lens]


def mutate(rng, seq, sub=0.10, indel=0.02, alphabet=DNA):
    """related-pair generator: substitutions and single-base indels."""
    out = bytearray()
    for c in seq:
        u = rng.random()
        if u < indel / 2:
            continue                                   # deletion
        if u < indel:
            out.append(int(alphabet[rng.integers(0, len(alphabet))]))   # insertion
        if rng.random() < sub:
            out.append(int(alphabet[rng.integers(0, len(alphabet))]))
        else:
            out.append(c)
    return bytes(out) if out else bytes(seq[:1])


def cigar_ops(text):
    return [(int(n), op) for n, op in re.findall(r"(\d+)([=XID])", text)]


def score_from_cigar(text, q, r, bq, br, scores, mapper, open_, ext):
    """Re-derive the alignment score from a CIGAR (I consumes the query, D the reference)."""
    i, j, s = bq, br, 0
    for n, op in cigar_ops(text):
        if op in "=X":
            for _ in range(n):
                s += int(scores[mapper[q[i]], mapper[r[j]]])
                i += 1; j += 1
        else:
            s -= open_ + (n - 1) * ext
            if op == "I":
                i += n
            else:
                j += n
    return s, i, j
