#!/usr/bin/env python3
"""bench.py -- BASELINE.json's configs on MI355X.

  python bench.py [--config 2|3|4|5] [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default: config 2, the headline -- a "step" is one pass of the hot path (`sw_striped_16`, score + end
positions) over one batch of synthetic pairs already resident in HBM: 1 000 000 pairs of 150 bp x 150 bp
i.i.d. DNA, Matrix::create("ACGT", 2, -3), gap open 5 / extend 2.
`--config 3|4|5` measures the other BASELINE configs the same way (workloads.py), each at ONE GPU's share of
the config (cfg 3: all 100k references -- the config is quoted on one GPU; cfg 4 / 5: 1.25M of the 10M).
Weak scaling (default): every rank holds its own batch of that size.  `--scaling strong`: the N = 1 batch is
split across the ranks by the shard planner (uniform for equal lengths, cumulative cells for mixed lengths).
No data-path collective; for N > 1 the records of each step are gathered to rank 0 over RCCL, overlapped
with the next step's kernel.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import workloads as wl  # noqa: E402

HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9           # 256 CUs x 4 SIMDs x 2.4 GHz

# Back-compat names (tests import them)
N_PAIRS = wl.CFG2["n"]
LEN = wl.CFG2["len"]


def make_cfg2_inputs(n=N_PAIRS, seed=wl.CFG2["seed"]):
    return wl.make_cfg2(n, rank=seed - wl.CFG2["seed"])


# VALU ceilings (DESIGN.md "Roofline").  Measured issue rates on this chip (profiles/r01/valu_rate_microbench.txt):
# a VOP3 / VOP3P wave64 instruction (v_pk_maximum3_f16, v_perm_b32, v_bfi_b32, packed adds) takes 4 cycles of its
# SIMD, a 32-bit-encoded VOP2 (v_add_u32, v_sub_u32) 2 cycles.  Counted inner-loop instructions per 2 cells x 64
# lanes (= 128 cells), by kernel family: (VOP3, VOP2, what).
VALU_MODEL = {
    "pmx_sw16_kernel": (4.5, 3.0, "sw16 skewed max3+vop2 variant: 4.5 VOP3/VOP3P + 3 VOP2 per 128 cells in steps that improve no lane's best "
                                  "(+1 v_bfi_b32 per 128 cells in the steps that do: the end-position strip save, data-dependent since round 2)"),
    "pmx_sw16q_kernel": (4.5, 3.0, "sw16q (shared profile): 4.5 VOP3/VOP3P + 3 VOP2 per 128 cells in steps that improve no lane's best (+1 in those that do)"),
    "pmx_stats16p_kernel": (30.0, 4.0, "stats16p: score arithmetic 7 + nine statistic planes moved by v_bfi_b32 under sign masks"),
    "pmx_stats16c_kernel": (17.0, 4.0, "stats16c: score arithmetic 7 + one combined statistics word per H/E/F moved by v_cndmask"),
    "pmx_nwsg16v_kernel/packed trace": (12.25, 2.0, "nwsg16v + traceback, row offset: score 4 VOP3 + 2 VOP2 (no F - extend), 4 packed differences + 3.5 v_bfi merges + 0.75 v_perm per 128 cells"),
    "pmx_nwsg16q_kernel/packed trace": (12.1, 2.0, "nwsg16q + traceback, row offset: score 4 VOP3 + 2 VOP2 (no F - extend), 4 packed differences + 3.5 v_bfi merges + 0.6 v_perm per 128 cells"),
    "pmx_nwsg16v_kernel": (4.0, 2.0, "nwsg16v, row offset: 4 VOP3/VOP3P + 2 VOP2 per 128 cells"),
}


class _Works:
    """Several collectives of one exchange step behind one wait()."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def valu_ceiling(kernel):
    for key in sorted(VALU_MODEL, key=len, reverse=True):
        if key.split("/")[0] in kernel and all(p in kernel for p in key.split("/")[1:]):
            v3, v2, what = VALU_MODEL[key]
            cyc = v3 * 4 + v2 * 2
            return SIMD_CYCLES_PER_S / cyc * 128.0 / 1e9, "%s = %.1f SIMD cycles; 1024 SIMDs x 2.4 GHz" % (what, cyc)
    return None, None


def valu_instructions_per_128(kernel):
    for key in sorted(VALU_MODEL, key=len, reverse=True):
        if key.split("/")[0] in kernel and all(p in kernel for p in key.split("/")[1:]):
            return VALU_MODEL[key][0] + VALU_MODEL[key][1]
    return None


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    cores = min(cores, max(1, q // per))
        except Exception:
            pass
    return cores


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def pmc_summary(config, kernel):
    """Counters of the dominant kernel from the committed PMC passes (profiles/r*/cfg<N>_pmc_summary.json, made by
    profiles/collect.sh: separate --pmc passes of the serialised pipeline).  Used only when the profiled kernel is the one that
    just ran AND its source file is byte-for-byte the one the counters were collected on (the summary records the git blob ids;
    a kernel edit that keeps the name must not report stale counters); the source file is named in the line.  gfx950
    correction (MI355X_MICROARCH.md): FETCH_SIZE counts half of the fetched bytes; both counters are in KiB."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cfg%d_pmc_summary.json" % config)))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        if d.get("kernel", "") and d["kernel"].split("<")[0] not in kernel:
            return None
        out = {"source": os.path.relpath(files[-1], ROOT), "profiled_kernel": d.get("kernel")}
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from summarize_pmc import source_blobs
        recorded = d.get("source_blobs")
        if not recorded:
            return {"source": out["source"], "stale": "%s records no source blob ids: counters not quoted" % out["source"]}
        now = source_blobs(d.get("kernel", ""))
        changed = [f for f in recorded if now.get(f) != recorded[f]]
        if changed:
            return {"source": out["source"],
                    "stale": "%s was collected on another version of %s (blob %s, now %s): counters not quoted"
                             % (out["source"], changed[0], recorded[changed[0]][:10], str(now.get(changed[0]))[:10])}
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            per_step = float(d.get("launches_per_step", 1.0))       # chunked pipelines launch the sweep several times per step
            out["traffic"] = int((2.0 * d["FETCH_SIZE"]["mean_per_launch"] + d["WRITE_SIZE"]["mean_per_launch"]) * 1024 * per_step)
            out["launches_per_step"] = per_step
        if "SQ_INSTS_VALU" in d and "GRBM_GUI_ACTIVE" in d:
            # measured issue rate of the dominant kernel (serialised launches): GRBM_GUI_ACTIVE sums the 8 XCDs, 1 024 SIMDs issue
            cyc = d["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
            valu = d["SQ_INSTS_VALU"]["mean_per_launch"]
            out["valu_issue"] = {"valu_instructions_per_launch": int(valu), "cycles_per_launch": int(cyc),
                                 "cycles_per_valu_instruction_per_simd": round(cyc / max(1.0, valu / 1024.0), 3),
                                 "note": "a VOP3 / VOP3P wave64 instruction occupies its SIMD for 4 cycles: at ~4 the kernel is bound by "
                                         "VALU issue and only fewer instructions make it faster"}
        if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
            out["lds_bank_conflict"] = {"conflict_cycles": int(d["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"]),
                                        "lds_active_cycles": int(d["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"]),
                                        "frac": round(d["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] /
                                                      max(1.0, d["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"]), 4)}
        return out
    except Exception as e:
        return {"source": os.path.relpath(files[-1], ROOT), "stale": "unreadable summary (%s)" % e}


# ------------------------------------------------------------------------------------------ workloads ----
class Workload:
    """One BASELINE config: inputs resident in HBM, one `step()` = one pass of the hot path over the batch."""
    config = 0
    dtype = "int16"
    default_steps, default_warmup = 20, 2

    def __init__(self, pkg, torch, dev, rank, world, scaling, n_override):
        self.pkg, self.torch, self.dev, self.rank, self.world, self.scaling = pkg, torch, dev, rank, world, scaling
        self.n_override = n_override

    def shard(self, qlens, rlens):
        """Strong scaling: this rank's [lo, hi) of the N = 1 batch and every rank's count."""
        from importlib import import_module
        sh = import_module("parasail_rs_amd.sharding")
        n = len(rlens)
        if np.all(rlens == rlens[0]) and np.all(qlens == qlens[0]):
            b = sh.shard_bounds_uniform(n, self.world)
        else:
            b = sh.shard_bounds_by_cells(qlens, rlens, self.world)
        return b[self.rank], b[self.rank + 1], [b[k + 1] - b[k] for k in range(self.world)]

    def to_dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def records(self, k):
        return self.d_out[k % 2]


class Cfg2(Workload):
    config = 2
    metric = "GCUPS (cell updates/s) local-affine SW, 1M 150x150 pairs, 1/2/4/8 GPUs"
    default_steps, default_warmup = 100, 5

    def setup(self):
        c = wl.CFG2
        n = self.n_override or c["n"]
        strong = self.scaling == "strong" and self.world > 1
        qbuf, qoff, rbuf, roff = wl.make_cfg2(n, rank=0 if strong else self.rank)
        self.counts = [n] * self.world
        if strong:
            lo, hi, self.counts = self.shard(np.full(n, c["len"]), np.full(n, c["len"]))
            qbuf, rbuf = qbuf[qoff[lo]:qoff[hi]], rbuf[roff[lo]:roff[hi]]
            qoff, roff = qoff[lo:hi + 1] - qoff[lo], roff[lo:hi + 1] - roff[lo]
            n = hi - lo
        self.n, self.h = n, (qbuf, qoff, rbuf, roff)
        self.d = [self.to_dev(x) for x in self.h]
        self.d_out = [self.torch.zeros((n, 4), dtype=self.torch.int32, device=self.dev) for _ in range(2)]
        self.matrix = self.pkg.Matrix.create(c["matrix"][0].encode(), c["matrix"][1], c["matrix"][2])
        self.cfg = self.pkg.pmx_config_t(self.pkg.MODE_SW, 0, c["open"], c["ext"], 16, 0, self.matrix.inner)
        self.cells = n * c["len"] * c["len"]
        self.algo_bytes = (2 * c["len"] + 12) * n          # 312 B / pair
        self.algo_note = "150 + 150 sequence bytes + 12 record bytes per pair (SURVEY.md 8d)"
        self.workload = "cfg2: %d pairs/GPU x (150 bp x 150 bp) i.i.d. DNA, %s (score + end positions), " \
                        "Matrix::create(ACGT,2,-3), gaps 5/2" % (n, c["name"])

    def step(self, k, stream):
        d = self.d
        self.pkg.align_batch_device(self.cfg, self.n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                    wl.CFG2["len"], wl.CFG2["len"], self.d_out[k % 2].data_ptr(), None, stream.cuda_stream)

    def pcie_inclusive(self):
        c = wl.CFG2
        al = self.pkg.Aligner.new().local().matrix(self.matrix).gap_open(c["open"]).gap_extend(c["ext"]).solution_width(16).build()
        out = np.zeros(self.n, dtype=self.pkg.RECORD_DTYPE)        # the caller's result buffer, reused from call to call

        def best(f):
            f(); ts = []
            for _ in range(5):
                t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
            return min(ts)
        t = best(lambda: al.align_batch_packed(*self.h, out=out))
        res = {"value": round(self.cells / t / 1e9, 1), "unit": "GCUPS", "ms": round(t * 1e3, 3),
               "entry": "pmx_align_batch (pageable host buffers in, host records out)"}
        ref = out.copy()
        try:                                   # the same with the caller's buffers page-locked once (pmx_host_register)
            self.pkg.host_register(*self.h, out)
            t = best(lambda: al.align_batch_packed(*self.h, out=out))
            res["pinned"] = {"value": round(self.cells / t / 1e9, 1), "unit": "GCUPS", "ms": round(t * 1e3, 3),
                             "entry": "pmx_align_batch, host buffers page-locked by the caller (pmx_host_register)"}
        finally:
            self.pkg.host_unregister(*self.h, out)
        # 2-bit packed input form (a quarter of the sequence bytes over PCIe, spelled out on the device)
        q2, r2 = self.pkg.pack_2bit(self.h[0]), self.pkg.pack_2bit(self.h[2])
        out[:] = 0
        t = best(lambda: al.align_batch_2bit(q2, self.h[1], r2, self.h[3], out=out))
        res["packed_2bit"] = {"value": round(self.cells / t / 1e9, 1), "unit": "GCUPS", "ms": round(t * 1e3, 3),
                              "entry": "pmx_align_batch_2bit (2 bits per base in pageable host buffers, host records out)",
                              "identical_records": bool((out == ref).all())}
        return res

    def cpu_baseline(self, last_out):
        from oracle import oracle as orc
        c = wl.CFG2
        m = orc.Matrix.create(c["matrix"][0], c["matrix"][1], c["matrix"][2])
        cores = usable_cores()
        qbuf, qoff, rbuf, roff = self.h
        L = c["len"]

        def run(npairs, lanes):
            t0 = time.perf_counter()
            out, used = orc.cpu_sw_striped16_batch(qbuf[: npairs * L], qoff[: npairs + 1], rbuf[: npairs * L],
                                                   roff[: npairs + 1], c["open"], c["ext"], m, threads=cores, lanes=lanes)
            return time.perf_counter() - t0, used, out
        widths = (16, 32) if orc.cpu_striped_lanes() == 32 else (16,)
        run(2048, 16)                                        # warm-up (thread pool, page-in)
        rate = {w: 16384 / max(run(16384, w)[0], 1e-6) for w in widths}
        lanes = max(rate, key=rate.get)                      # the faster vector width on this CPU
        sample = int(min(self.n, max(16384, rate[lanes] * 2.0)))     # ~2 s wall
        t, used, out = run(sample, lanes)
        v = sample * L * L / t / 1e9
        agrees = bool((last_out[:sample, :3] == out).all())
        return {"value": round(v, 3), "unit": "GCUPS", "cores": int(used), "per_core": round(v / used, 3), "kind": "port",
                "sample": "%d of the same 150x150 pairs, restated CPU baseline (not parasail): striped int16 %s + OpenMP, "
                          "per-pair profiles (oracle/pmx_striped_cpu.c; the faster of the CPU's vector widths), %s, %.2f s wall"
                          % (sample, "AVX-512BW x32" if lanes == 32 else "AVX2 x16", cpu_model(), t),
                "agrees_with_gpu": agrees}


def _profile_host_entry(w, stats):
    import ctypes as C
    pkg, (rbuf, roff) = w.pkg, w.h
    out = np.zeros(w.n, dtype=pkg.RECORD_DTYPE)
    st = np.zeros(w.n, dtype=pkg.STATS_DTYPE) if stats else None

    def call():
        rc = pkg.lib.pmx_align_profile_batch(C.byref(w.cfg), w.profile.inner, w.n, rbuf.ctypes.data, roff.ctypes.data,
                                             out.ctypes.data, st.ctypes.data if stats else None)
        if rc:
            raise RuntimeError(pkg.lib.pmx_last_error().decode())
    call()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    dev = w.records(w.last_k).cpu().numpy()
    same = bool((out["score"] == dev[:, 0]).all() and (out["end_query"] == dev[:, 1]).all() and (out["end_ref"] == dev[:, 2]).all())
    return {"value": round(w.cells / min(ts) / 1e9, 1), "unit": "GCUPS", "ms": round(min(ts) * 1e3, 3),
            "entry": "pmx_align_profile_batch (pageable host references in, host records%s out; references go up in slices behind the "
                     "kernels)" % (" + statistics" if stats else ""),
            "identical_records": same}


class Cfg3(Workload):
    config = 3
    metric = "GCUPS (cell updates/s) global NW + statistics, reused 300-aa profile vs 100k x ~4.75 kaa, BLOSUM62 11/1"
    default_steps, default_warmup = 10, 2

    def setup(self):
        c = wl.CFG3
        n = self.n_override or c["n"]
        strong = self.scaling == "strong" and self.world > 1
        q, rbuf, roff = wl.make_cfg3(n, rank=0 if strong else self.rank)
        self.counts = [n] * self.world
        if strong:
            rl = roff[1:] - roff[:-1]
            lo, hi, self.counts = self.shard(np.full(n, c["qlen"]), rl)
            rbuf, roff = rbuf[roff[lo]:roff[hi]], roff[lo:hi + 1] - roff[lo]
            n = hi - lo
        self.n, self.q, self.h = n, q, (rbuf, roff)
        self.d = [self.to_dev(x) for x in self.h]
        self.d_out = [self.torch.zeros((n, 4), dtype=self.torch.int32, device=self.dev) for _ in range(2)]
        self.d_st = [self.torch.zeros((n, 3), dtype=self.torch.int32, device=self.dev) for _ in range(2)]
        self.matrix = self.pkg.Matrix.from_name(c["matrix"])
        self.profile = self.pkg.Profile.new(q, True, self.matrix)
        self.cfg = self.pkg.pmx_config_t(self.pkg.MODE_NW, 0, c["open"], c["ext"], 16, self.pkg.WANT_STATS, self.matrix.inner)
        self.max_rlen = int((roff[1:] - roff[:-1]).max())
        self.cells = int(c["qlen"] * roff[-1])
        self.algo_bytes = int(roff[-1]) + 24 * n          # rlen + 24 B / pair
        self.algo_note = "reference bytes + 12 record + 12 statistics bytes per pair (SURVEY.md 8d); the profile is read once per workgroup"
        self.workload = "cfg3: one 300-aa query (reused stats profile) x %d references/GPU of 4.5-5 kaa, %s " \
                        "(score, ends, matches, similar, length), BLOSUM62, gaps 11/1" % (n, c["name"])

    def step(self, k, stream):
        self.pkg.align_profile_batch_device(self.cfg, self.profile, self.n, self.d[0].data_ptr(), self.d[1].data_ptr(),
                                            self.max_rlen, self.d_out[k % 2].data_ptr(), self.d_st[k % 2].data_ptr(),
                                            stream.cuda_stream)

    def pcie_inclusive(self):
        """the same batch through pmx_align_profile_batch: references from host memory, records + statistics back to host memory"""
        return _profile_host_entry(self, stats=True)

    def cpu_baseline(self, last_out):
        from oracle import oracle as orc
        c = wl.CFG3
        om = orc.Matrix.from_file(os.path.join(ROOT, "tests", "golden", "blosum62.txt"))
        cores = usable_cores()
        os.environ["OMP_NUM_THREADS"] = str(cores)
        rbuf, roff = self.h
        # vectorised port (oracle/pmx_cpu_inter16.c: 16 references per AVX2 vector, every lane the oracle's comparisons in the oracle's
        # order), checked against the scalar oracle on the first references, then timed on a bounded sample
        m0 = min(self.n, 2 * cores)
        want0 = orc.align_stats_sample(orc.NW, np.arange(m0), None, None, rbuf, roff, c["open"], c["ext"], om, bits=16, shared_query=self.q)
        warm, _ = orc.cpu_nw_stats_inter16(self.q, rbuf[: roff[m0]], roff[: m0 + 1], c["open"], c["ext"], om, threads=cores)
        port_ok = bool((warm == want0[:, :6]).all())
        t0 = time.perf_counter()
        probe, used = orc.cpu_nw_stats_inter16(self.q, rbuf[: roff[16 * cores]], roff[: 16 * cores + 1], c["open"], c["ext"], om, threads=cores)
        rate = 16 * cores / max(time.perf_counter() - t0, 1e-6)
        m = int(min(self.n, max(16 * cores, (rate * 10.0) // (16 * cores) * (16 * cores))))      # ~10 s wall, whole groups per thread
        t0 = time.perf_counter()
        got, used = orc.cpu_nw_stats_inter16(self.q, rbuf[: roff[m]], roff[: m + 1], c["open"], c["ext"], om, threads=cores)
        t = time.perf_counter() - t0
        v = c["qlen"] * int(roff[m]) / t / 1e9
        st = self.d_st[self.last_k % 2][:m].cpu().numpy()
        agrees = bool(port_ok and (last_out[:m, :3] == got[:, :3]).all() and (st == got[:, 3:6]).all())
        return {"value": round(v, 3), "unit": "GCUPS", "cores": int(used), "per_core": round(v / used, 3), "kind": "port",
                "sample": "the first %d references of the same batch, restated CPU baseline (not parasail): global alignment with the "
                          "coupled statistics in int16 AVX2, 16 references per vector (inter-sequence: no lazy-F pass; "
                          "oracle/pmx_cpu_inter16.c, equal to the scalar oracle on the first %d references) + OpenMP, the query "
                          "profile of a column built once per 16 references; %s, %.2f s wall" % (m, m0, cpu_model(), t),
                "agrees_with_gpu": agrees}


class Cfg4(Workload):
    config = 4
    metric = "GCUPS (cell updates/s) semi-global affine with traceback + CIGAR text, 250x250 related DNA pairs"
    default_steps, default_warmup = 20, 2
    TEXT_CAP_PER_PAIR = 320           # bytes of CIGAR text reserved per pair (the entry reports the bytes it needed)

    def setup(self):
        c = wl.CFG4
        n = self.n_override or c["n"] // 8
        strong = self.scaling == "strong" and self.world > 1
        qbuf, qoff, rbuf, roff = wl.make_cfg4(n, rank=0 if strong else self.rank)
        self.counts = [n] * self.world
        if strong:
            lo, hi, self.counts = self.shard(np.full(n, c["len"]), np.full(n, c["len"]))
            qbuf, rbuf = qbuf[qoff[lo]:qoff[hi]], rbuf[roff[lo]:roff[hi]]
            qoff, roff = qoff[lo:hi + 1] - qoff[lo], roff[lo:hi + 1] - roff[lo]
            n = hi - lo
        self.n, self.h = n, (qbuf, qoff, rbuf, roff)
        self.d = [self.to_dev(x) for x in self.h]
        t = self.torch
        self.d_out = [t.zeros((n, 4), dtype=t.int32, device=self.dev) for _ in range(2)]
        self.cap = self.TEXT_CAP_PER_PAIR * n
        self.d_text = [t.zeros(self.cap, dtype=t.uint8, device=self.dev) for _ in range(2)]
        self.d_toff = [t.zeros(n + 1, dtype=t.int64, device=self.dev) for _ in range(2)]
        self.matrix = self.pkg.Matrix.create(c["matrix"][0].encode(), c["matrix"][1], c["matrix"][2])
        self.cfg = self.pkg.pmx_config_t(self.pkg.MODE_SG, self.pkg.SG_ALL, c["open"], c["ext"], 16, self.pkg.WANT_CIGAR, self.matrix.inner)
        self.cells = n * c["len"] * c["len"]
        self.algo_bytes = None                         # 512 B/pair + the CIGAR text actually produced: known after the first step
        self.algo_note = "250 + 250 sequence bytes + 12 record bytes + the CIGAR text produced per pair (SURVEY.md 8d); the 4-bit " \
                         "traceback cells are scratch, not algorithmic bytes"
        self.workload = "cfg4: %d pairs/GPU x (250 bp x 250 bp) related DNA (10%% subs, 2%% indels), %s + CIGAR text on the " \
                        "device, Matrix::create(ACGT,2,-3), gaps 5/2" % (n, c["name"])

    def step(self, k, stream):
        d = self.d
        self.pkg.align_batch_cigar_device(self.cfg, self.n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                          wl.CFG4["len"], wl.CFG4["len"], self.d_out[k % 2].data_ptr(),
                                          self.d_text[k % 2].data_ptr(), self.cap, self.d_toff[k % 2].data_ptr(), stream.cuda_stream)

    def exchange(self, k, sharding, backend, cache=None):
        """N > 1: this step's CIGAR text follows the records to rank 0 (sizes first, then offsets and text padded to the longest
        shard; parasail-rs_amd/sharding.py:gather_text).  Returns the works still in flight."""
        text, toff = self.d_text[k % 2], self.d_toff[k % 2]
        if backend != "nccl":
            text, toff = text.cpu(), toff.cpu()
        fin, works = sharding.gather_text(text, toff, self.counts, dst=0, async_op=True, cache=cache)
        self.last_text_gather = fin
        return works

    def check_exchange(self, last_records):
        """one-rank rehearsal of the exchange: the gathered records, text and offsets equal the local ones"""
        k = self.last_k
        text, toff = self.last_text_gather()
        n_bytes = int(self.d_toff[k % 2][-1].item())
        return {"records_equal_local": bool((last_records().cpu() == self.d_out[k % 2].cpu()).all().item()),
                "text_equal_local": bool((text.cpu() == self.d_text[k % 2][:n_bytes].cpu()).all().item()),
                "offsets_equal_local": bool((toff.cpu() == self.d_toff[k % 2].cpu()).all().item()),
                "text_bytes": n_bytes}

    def finish(self):
        total = int(self.d_toff[self.last_k % 2][-1].item())
        if total > self.cap:
            raise RuntimeError("CIGAR text capacity too small: %d > %d" % (total, self.cap))
        self.text_bytes = total
        self.algo_bytes = (2 * wl.CFG4["len"] + 12) * self.n + total

    def pcie_inclusive(self):
        c = wl.CFG4
        al = self.pkg.Aligner.new().semi_global().matrix(self.matrix).gap_open(c["open"]).gap_extend(c["ext"]).solution_width(16) \
            .use_trace().build()
        out = np.zeros(self.n, dtype=self.pkg.RECORD_DTYPE); coff = np.zeros(self.n + 1, dtype=np.int64)     # the caller's result arrays, reused
        al.align_batch_cigar_packed(*self.h, out=out, coff=coff)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); al.align_batch_cigar_packed(*self.h, out=out, coff=coff); ts.append(time.perf_counter() - t0)
        return {"value": round(self.cells / min(ts) / 1e9, 1), "unit": "GCUPS", "ms": round(min(ts) * 1e3, 3),
                "entry": "pmx_align_batch_cigar (host buffers in, host records + CIGAR text out; result arrays reused, text block recycled by pmx_free)"}

    def cpu_baseline(self, last_out):
        from oracle import oracle as orc
        c = wl.CFG4
        om = orc.Matrix.create(c["matrix"][0], c["matrix"][1], c["matrix"][2])
        cores = usable_cores()
        os.environ["OMP_NUM_THREADS"] = str(cores)
        qbuf, qoff, rbuf, roff = self.h
        # vectorised port (oracle/pmx_cpu_inter16.c: 16 pairs per AVX2 vector, byte trace table, scalar walk, CIGAR text), checked
        # against the scalar oracle on the first pairs, then timed on a bounded sample
        m0 = min(self.n, 16 * cores)
        text0, rec0 = orc.cigar_sample(orc.SG, np.arange(m0), qbuf, qoff, rbuf, roff, c["open"], c["ext"], om)
        t0_, r0_, _ = orc.cpu_trace_cigar_inter16(orc.SG, qbuf[: qoff[m0]], qoff[: m0 + 1], rbuf[: roff[m0]], roff[: m0 + 1], c["open"], c["ext"], om, threads=cores)
        port_ok = bool((r0_ == rec0).all() and [x.decode() for x in t0_] == text0)
        m1 = min(self.n, 256 * cores)
        t0 = time.perf_counter()
        orc.cpu_trace_cigar_inter16(orc.SG, qbuf[: qoff[m1]], qoff[: m1 + 1], rbuf[: roff[m1]], roff[: m1 + 1], c["open"], c["ext"], om, threads=cores, decode=False)
        rate = m1 / max(time.perf_counter() - t0, 1e-6)
        m = int(min(self.n, max(m1, (rate * 10.0) // (16 * cores) * (16 * cores))))                 # ~10 s wall
        t0 = time.perf_counter()
        text, rec, used = orc.cpu_trace_cigar_inter16(orc.SG, qbuf[: qoff[m]], qoff[: m + 1], rbuf[: roff[m]], roff[: m + 1], c["open"], c["ext"], om, threads=cores, decode=False)
        t = time.perf_counter() - t0
        v = m * c["len"] * c["len"] / t / 1e9
        toff = self.d_toff[self.last_k % 2][: m + 1].cpu().numpy()
        raw = self.d_text[self.last_k % 2][: int(toff[-1])].cpu().numpy().tobytes()
        blob, stride = text.tobytes(), text.shape[1]                    # NUL-terminated slots of the port's text
        agrees = bool(port_ok and (last_out[:m, :3] == rec[:, :3]).all() and
                      all(blob[k * stride:k * stride + int(toff[k + 1] - toff[k]) + 1] == raw[toff[k]:toff[k + 1]] + b"\0" for k in range(m)))
        return {"value": round(v, 3), "unit": "GCUPS", "cores": int(used), "per_core": round(v / used, 3), "kind": "port",
                "sample": "the first %d pairs of the same batch, restated CPU baseline (not parasail): semi-global alignment with the byte "
                          "trace table in int16 AVX2, 16 pairs per vector (inter-sequence: no lazy-F pass), scalar walk + CIGAR text per "
                          "pair (oracle/pmx_cpu_inter16.c, equal to the scalar oracle on the first %d pairs) + OpenMP; %s, %.2f s wall"
                          % (m, m0, cpu_model(), t),
                "agrees_with_gpu": agrees}


class Cfg5(Workload):
    config = 5
    metric = "GCUPS (cell updates/s) local SW, reused 1 kbp profile vs 0.5-5 kbp references, sat (8->16->32 promotion)"
    default_steps, default_warmup = 5, 1

    def setup(self):
        c = wl.CFG5
        n = self.n_override or c["n"] // 8
        strong = self.scaling == "strong" and self.world > 1
        q, rbuf, roff, planted = wl.make_cfg5(n, rank=0 if strong else self.rank)
        self.counts = [n] * self.world
        if strong:
            rl = roff[1:] - roff[:-1]
            lo, hi, self.counts = self.shard(np.full(n, c["qlen"]), rl)
            rbuf, roff = rbuf[roff[lo]:roff[hi]], roff[lo:hi + 1] - roff[lo]
            n = hi - lo
        self.n, self.q, self.h = n, q, (rbuf, roff)
        self.d = [self.to_dev(x) for x in self.h]
        self.d_out = [self.torch.zeros((n, 4), dtype=self.torch.int32, device=self.dev) for _ in range(2)]
        self.matrix = self.pkg.Matrix.create(c["matrix"][0].encode(), c["matrix"][1], c["matrix"][2])
        self.profile = self.pkg.Profile.new(q, False, self.matrix)
        # width 0 = sat; PMX_WANT_SORTED: mixed lengths are processed in length-sorted order (records stay in input order)
        self.cfg = self.pkg.pmx_config_t(self.pkg.MODE_SW, 0, c["open"], c["ext"], 0, self.pkg.WANT_SORTED, self.matrix.inner)
        self.max_rlen = int((roff[1:] - roff[:-1]).max())
        self.cells = int(c["qlen"] * roff[-1])
        self.algo_bytes = int(roff[-1]) + 12 * n
        self.algo_note = "reference bytes + 12 record bytes per pair (SURVEY.md 8d); the 1 kbp query is read once per workgroup"
        self.workload = "cfg5: one 1 kbp query (reused profile) x %d references/GPU of 0.5-5 kbp (log-uniform; 1%% carry a noisy " \
                        "copy), %s (8->16->32 promotion, none needed beyond int16 here), Matrix::create(ACGT,2,-3), gaps 5/2; " \
                        "length sort included in the step" % (n, c["name"])

    def step(self, k, stream):
        self.pkg.align_profile_batch_device(self.cfg, self.profile, self.n, self.d[0].data_ptr(), self.d[1].data_ptr(),
                                            self.max_rlen, self.d_out[k % 2].data_ptr(), None, stream.cuda_stream)

    def pcie_inclusive(self):
        return _profile_host_entry(self, stats=False)

    def extra(self):
        """The "banded SW" arm of config 5 (an extension: the reference has banded_nw only): a second pass restricted to
        |(j - i) - diag| <= band around the diagonal of the first pass's end cell, band-only kernel (pmx_banded.hip)."""
        t = self.torch
        band = 48
        rec = self.d_out[self.last_k % 2]
        d_diag = (rec[:, 2] - rec[:, 1]).contiguous()
        d_out = t.zeros((self.n, 4), dtype=t.int32, device=self.dev)
        stream = t.cuda.current_stream(self.dev)
        cfg = self.pkg.pmx_config_t(self.pkg.MODE_SW, 0, wl.CFG5["open"], wl.CFG5["ext"], 0, 0, self.matrix.inner)

        def once():
            rc = self.pkg.lib.pmx_align_profile_batch_banded_device(C.byref(cfg), self.profile.inner, self.n, self.d[0].data_ptr(),
                                                                    self.d[1].data_ptr(), self.max_rlen, band, d_diag.data_ptr(),
                                                                    d_out.data_ptr(), stream.cuda_stream)
            if rc:
                raise RuntimeError(self.pkg.lib.pmx_last_error().decode())
        once(); t.cuda.synchronize(self.dev)
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        reps = 3
        e0.record(stream)
        for _ in range(reps):
            once()
        e1.record(stream); t.cuda.synchronize(self.dev)
        ms = e0.elapsed_time(e1) / reps
        rl = (self.h[1][1:] - self.h[1][:-1]).astype(np.int64)
        dg = d_diag.cpu().numpy().astype(np.int64)
        ql = wl.CFG5["qlen"]
        # band cells of a pair: for every query row i the columns j in [i + diag - band, i + diag + band] inside [0, rlen)
        i = np.arange(ql, dtype=np.int64)[None, :]
        sample = np.arange(0, self.n, max(1, self.n // 4096))
        lo = np.maximum(i + dg[sample, None] - band, 0); hi = np.minimum(i + dg[sample, None] + band, rl[sample, None] - 1)
        band_cells = float(np.maximum(hi - lo + 1, 0).sum()) * (self.n / len(sample))
        same = bool((d_out[:, 0] <= rec[:, 0]).all().item())
        kept = float((d_out[:, 0] == rec[:, 0]).float().mean().item())
        return {"banded_sw": {"band": band, "ms": round(ms, 3), "pairs_per_s": round(self.n / (ms * 1e-3)),
                              "band_gcups": round(band_cells / (ms * 1e-3) / 1e9, 1),
                              "band_cells_inside_the_matrix": int(band_cells),
                              "kernel": self.pkg.lib.pmx_last_kernel().decode(),
                              "never_above_full_pass": same, "share_of_pairs_with_the_full_score": round(kept, 4),
                              "note": "extension, no reference counterpart: |(j - i) - diag| <= band with diag = end_ref - end_query of "
                                      "the first pass; only the band's cells are computed and only they are counted (band-strip kernel: band "
                                      "coordinates, 13 offsets per lane, packed int16, two pairs per lane group; the sort by band length and "
                                      "the second launch that settles end-cell ties are inside the time)"}}

    def cpu_baseline(self, last_out):
        from oracle import oracle as orc
        c = wl.CFG5
        om = orc.Matrix.create(c["matrix"][0], c["matrix"][1], c["matrix"][2])
        cores = usable_cores()
        rbuf, roff = self.h

        def run(m, lanes):
            t0 = time.perf_counter()
            out, used = orc.cpu_sw_striped16_batch(None, None, rbuf[: roff[m]], roff[: m + 1], c["open"], c["ext"], om,
                                                   threads=cores, shared_query=self.q, lanes=lanes)
            return time.perf_counter() - t0, used, out
        widths = (16, 32) if orc.cpu_striped_lanes() == 32 else (16,)
        run(min(self.n, 4 * cores), 16)
        m0 = min(self.n, 64 * cores)
        rate = {w: m0 / max(run(m0, w)[0], 1e-6) for w in widths}
        lanes = max(rate, key=rate.get)
        m = int(min(self.n, max(m0, rate[lanes] * 3.0)))
        t, used, out = run(m, lanes)
        v = c["qlen"] * int(roff[m]) / t / 1e9
        agrees = bool((last_out[:m, :3] == out).all())
        return {"value": round(v, 3), "unit": "GCUPS", "cores": int(used), "per_core": round(v / used, 3), "kind": "port",
                "sample": "the first %d references of the same batch, restated CPU baseline (not parasail): striped int16 %s + OpenMP, "
                          "the query profile built once per thread (oracle/pmx_striped_cpu.c), %s, %.2f s wall"
                          % (m, "AVX-512BW x32" if lanes == 32 else "AVX2 x16", cpu_model(), t),
                "agrees_with_gpu": agrees}


WORKLOADS = {2: Cfg2, 3: Cfg3, 4: Cfg4, 5: Cfg5}


class _Env:
    """What every config's run shares: the package, the device, the process group."""
    pass


def run_config(env, config, steps, warmup, scaling, pairs, cpu_legs=True, deferred=None):
    """One BASELINE config: W untimed + K timed steps bracketed by barrier + synchronize, max over ranks.
    Returns the JSON line as a dict on rank 0, None elsewhere.  deferred (a list): the host-side legs (`pcie_inclusive`,
    `cpu_baseline`) are appended to it as closures instead of being run here -- the default run measures every config on the GPU
    first and the CPU legs afterwards: a CPU leg of one config (16 OpenMP threads, pinned staging buffers) ahead of the next
    config's GPU steps cost those 4-5 % (their pipelines are many launches and stream waits per step)."""
    torch, pkg, dev, dist = env.torch, env.pkg, env.dev, env.dist
    rank, world, multi, backend, sharding = env.rank, env.world, env.multi, env.backend, env.sharding
    w = WORKLOADS[config](pkg, torch, dev, rank, world, scaling, pairs)
    w.setup()
    steps = steps if steps is not None else w.default_steps
    warmup = warmup if warmup is not None else w.default_warmup
    comm_stream = env.comm_stream
    pending = []
    recv_cache = [{}, {}]             # receive buffers of the exchange step, one set per output slot, allocated once

    def step(k, events=None):
        out = w.records(k)
        stream = torch.cuda.current_stream(dev)
        while len(pending) >= 2:          # the gather that last read this output buffer (and wrote this slot's receive buffers) must be done
            fin, work = pending.pop(0)
            work.wait()
        if events is not None:
            events[0].record(stream)
        w.step(k, stream)
        w.last_k = k
        if events is not None:
            events[1].record(stream)
        if multi:
            # exchange step: records of this step go to rank 0 while the next step computes
            cache = recv_cache[k % 2]
            if backend == "nccl":
                done = torch.cuda.Event()
                done.record(stream)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(done)
                    fin, work = sharding.gather_records(out, w.counts, dst=0, async_op=True, cache=cache)
                    extra = w.exchange(k, sharding, backend, cache) if hasattr(w, "exchange") else []
            else:
                fin, work = sharding.gather_records(out.cpu(), w.counts, dst=0, async_op=True, cache=cache)
                extra = w.exchange(k, sharding, backend, cache) if hasattr(w, "exchange") else []
            pending.append((fin, _Works([work] + list(extra))))
            env.last_gather = fin

    def drain():
        while pending:
            fin, work = pending.pop(0)
            work.wait()
        if comm_stream is not None:
            torch.cuda.current_stream(dev).wait_stream(comm_stream)

    for k in range(warmup):
        step(k)
    drain()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        step(k, evs[k])
    drain()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    cells_all = float(w.cells)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cells_all], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells_all = float(c.item())
    else:
        cells_all *= world
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if evs else float("nan")
    kernel = pkg.lib.pmx_last_kernel().decode()        # name + shape + arithmetic variant of what this thread just launched
    if hasattr(w, "finish"):
        w.finish()

    line = None
    if rank == 0:
        gcups = cells_all * steps / elapsed / 1e9
        achieved = w.algo_bytes / (kern_ms * 1e-3) / 1e9
        kern_gcups = w.cells / (kern_ms * 1e-3) / 1e9
        pmc = pmc_summary(config, kernel) if pairs is None and scaling == "weak" else None
        line = {
            "metric": w.metric, "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": w.dtype, "data": "synthetic",
            "value_is": "device-resident batch throughput (inputs already in HBM when the timed region starts, the bench contract's "
                        "definition); SURVEY.md 8(d)'s end-to-end form (H2D + kernels + D2H through the host entry) is `pcie_inclusive`",
            "config": {"workload": w.workload, "pairs_per_gpu": w.n, "kernel": kernel, "inputs": "resident in HBM",
                       "exchange": "none" if not multi else "%s gather of 16-B records%s to rank 0, overlapped" %
                                   ("RCCL" if backend == "nccl" else backend, " and of the CIGAR text" if hasattr(w, "exchange") else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc.get("traffic") if pmc else None,
                         "traffic_source": pmc.get("source") if pmc else None,
                         "traffic_over_algorithmic": round(pmc["traffic"] / w.algo_bytes, 2) if pmc and pmc.get("traffic") else None,
                         "kernel_ms": round(kern_ms, 4), "kernel_ms_clock": "HIP events around each step on the launch stream, inside bench.py "
                                                                           "(the rocprofv3 average of the same kernel under profiles/ is ~4 % longer: profiler overhead)",
                         "algorithmic_bytes": int(w.algo_bytes),
                         "note": w.algo_note + "; the path is VALU-bound, see roofline_valu"},
        }
        if pmc and "lds_bank_conflict" in pmc:
            line["lds_bank_conflict"] = dict(pmc["lds_bank_conflict"], source=pmc["source"])
        if pmc and pmc.get("stale"):
            line["roofline"]["traffic_note"] = pmc["stale"]
        peak, model = valu_ceiling(kernel)
        if peak:
            line["roofline_valu"] = {"bound": "valu", "achieved": round(kern_gcups, 2), "peak": round(peak, 1), "unit": "GCUPS",
                                     "frac": round(kern_gcups / peak, 4), "model": model}
            if pmc and "valu_issue" in pmc:
                line["roofline_valu"]["measured"] = dict(pmc["valu_issue"], source=pmc["source"])
                # the 4 / 2-cycle model above is optimistic for this VOP3P / VOP2 mix (profiles/r01/valu_rate_microbench.txt: a 1:1 mix
                # issues at ~3.9 cycles per instruction): the same instruction count at the issue rate the counters measured
                v3v2 = valu_instructions_per_128(kernel)
                if v3v2:
                    cpi = pmc["valu_issue"]["cycles_per_valu_instruction_per_simd"]
                    pk = SIMD_CYCLES_PER_S / (v3v2 * cpi) * 128.0 / 1e9
                    line["roofline_valu"]["peak_at_measured_issue_rate"] = round(pk, 1)
                    line["roofline_valu"]["frac_at_measured_issue_rate"] = round(kern_gcups / pk, 4)
        if multi and world == 1:
            # rehearsal of the exchange path with one rank (PMX_BENCH_FORCE_DIST): what rank 0 gathered must be what it computed
            line["exchange_check"] = w.check_exchange(env.last_gather) if hasattr(w, "check_exchange") else \
                {"records_equal_local": bool((env.last_gather().cpu() == w.records(w.last_k).cpu()).all().item())}
        if world == 1 and not multi and hasattr(w, "extra"):
            line.update(w.extra())
        last_out = w.records(w.last_k).cpu().numpy() if (world == 1 and cpu_legs) else None

        def host_legs(w=w, line=line, last_out=last_out):
            if world == 1 and not multi and pairs is None and cpu_legs and hasattr(w, "pcie_inclusive"):
                # the same batch handed over in host memory (H2D + kernels + D2H inside): reported beside, never as `value`
                line["pcie_inclusive"] = w.pcie_inclusive()
            if world == 1 and cpu_legs:
                line["cpu_baseline"] = w.cpu_baseline(last_out)
        if deferred is not None:
            # the host legs run after every config's GPU steps: they need the host inputs only -- the device buffers go now, so that the
            # next config is timed in the memory state `--config N` alone would see (the library sizes its chunks from hipMemGetInfo)
            for name, val in list(vars(w).items()):             # inputs (`d`) are dropped, the last outputs move to host memory
                if torch.is_tensor(val) and val.is_cuda:
                    setattr(w, name, None if name == "d" else val.cpu())
                elif isinstance(val, (list, tuple)) and val and all(torch.is_tensor(x) and x.is_cuda for x in val):
                    setattr(w, name, None if name == "d" else [x.cpu() for x in val])
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            deferred.append(host_legs)
            return line
        host_legs()
        del host_legs
    del w
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return line


# steps / warm-up of the configs that ride along with the default (headline) run: sized so that the whole default run --
# input generation included -- stays within a couple of minutes
ALONGSIDE = {3: (5, 1), 4: (10, 2), 5: (3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=sorted(WORKLOADS),
                    help="one config only (default: config 2 as the line's value, and at one GPU configs 3, 4, 5 beside it under \"configs\")")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--pairs", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--headline-only", action="store_true", help="default run without configs 3-5")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_pkg()

    env = _Env()
    env.torch, env.pkg = torch, pkg
    env.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        print("bench.py: --gpus %d needs torch.distributed.run with %d processes" % (args.gpus, args.gpus),
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # PMX_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share
    # devices, records are gathered through host memory); the driver's runs use nccl (= RCCL).
    env.backend = backend = os.environ.get("PMX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    env.dev = dev = torch.device("cuda", dev_index)
    env.dist = dist = None
    # PMX_BENCH_FORCE_DIST=1 (rehearsal on a one-GPU box): take the exchange path with a single rank
    env.multi = multi = world > 1 or bool(os.environ.get("PMX_BENCH_FORCE_DIST"))
    if multi:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        env.dist = dist
    from importlib import import_module
    env.sharding = import_module("parasail_rs_amd.sharding")
    env.comm_stream = torch.cuda.Stream(device=dev) if multi else None
    env.last_gather = None

    head = args.config if args.config is not None else 2
    alongside = (args.config is None and not args.headline_only and world == 1 and not multi and args.pairs is None
                 and args.scaling == "weak")
    deferred = [] if alongside else None      # (one config alone: its host legs follow its GPU steps at once)
    line = run_config(env, head, args.steps, args.warmup, args.scaling, args.pairs, cpu_legs=not args.no_cpu_baseline, deferred=deferred)
    if alongside:
        # the other BASELINE configs under the same clock (each: value, ms_per_step, roofline, roofline_valu, cpu_baseline, pcie_inclusive);
        # every config's GPU steps first, the host-side legs of all of them afterwards
        line["configs"] = {}
        for c in sorted(ALONGSIDE):
            t0 = time.perf_counter()
            sub = run_config(env, c, ALONGSIDE[c][0], ALONGSIDE[c][1], "weak", None, cpu_legs=not args.no_cpu_baseline, deferred=deferred)
            for key in ("higher_is_better", "vs_baseline", "data", "value_is", "n_gpus", "scaling"):
                sub.pop(key, None)
            sub["wall_s_gpu_part_incl_input_generation"] = round(time.perf_counter() - t0, 1)
            line["configs"][str(c)] = sub
        while deferred:
            deferred.pop(0)()
            import gc
            gc.collect()
    if rank == 0:
        if line.get("configs"):
            # the numbers of every config once more at the END of the line (a log tail keeps the end)
            line["summary"] = {"2": {"gcups": line["value"], "ms_per_step": line["ms_per_step"]}}
            for c, sub in line["configs"].items():
                line["summary"][c] = {"gcups": sub.get("value"), "ms_per_step": sub.get("ms_per_step")}
                if "banded_sw" in sub:
                    line["summary"][c]["banded_second_pass_ms"] = sub["banded_sw"]["ms"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
