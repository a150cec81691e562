#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (`sw_striped_16`, score + end positions) over one batch of
synthetic pairs that is already resident in HBM: BASELINE config 2, 1 000 000 pairs of 150 bp x
150 bp i.i.d. DNA, Matrix::create("ACGT", 2, -3), gap open 5 / extend 2.
Every rank holds its own 1M-pair batch (weak scaling, no data-path collective); for N > 1 the
records of each step are gathered to rank 0 over RCCL, overlapped with the next step's kernel.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PAIRS = 1_000_000
LEN = 150
SEED = 20260001
MATCH, MISMATCH, OPEN, EXT = 2, -3, 5, 2
ALGO_BYTES_PER_PAIR = LEN + LEN + 12          # 312 B / pair
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU ceiling for the hot kernel (DESIGN.md "Roofline").  Measured issue rates on this chip
# (profiles/r01/valu_rate_microbench.txt): a VOP3/VOP3P wave64 instruction (v_pk_maximum3_f16,
# v_perm_b32, v_bfi_b32) takes 4 cycles of its SIMD, a 32-bit-encoded VOP2 (v_add_u32, v_sub_u32)
# 2 cycles.  Per 2 cells x 64 lanes the skewed max3+vop2 variant issues 5.5 VOP3 + 3 VOP2 = 28 cycles.
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
CYCLES_PER_128_CELLS = 5.5 * 4 + 3 * 2


def make_cfg2_inputs(n=N_PAIRS, seed=SEED):
    """numpy.random.default_rng(seed): queries first, then references, 0..3 -> ACGT."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    q = lut[rng.integers(0, 4, size=(n, LEN), dtype=np.uint8)].reshape(-1)
    r = lut[rng.integers(0, 4, size=(n, LEN), dtype=np.uint8)].reshape(-1)
    off = np.arange(n + 1, dtype=np.int64) * LEN
    return q, off, r, off.copy()


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


def pmc_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/<round>/*pmc_summary.json: FETCH_SIZE and WRITE_SIZE in KiB, separate passes).
    gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE counts half of the fetched bytes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return int((2.0 * d["FETCH_SIZE"]["mean_per_launch"] + d["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    except Exception:
        return None


def cpu_reference_baseline(qbuf, rbuf, cores):
    """If the box has the reference's real library (a system libparasail), time `parasail_sw_striped_16`
    on all usable cores, one thread per core over its slice of pairs (the pattern of the reference's
    tests/test_parasail.rs:702-717).  Returns None when it is absent (it is, in the build image)."""
    try:
        from oracle import parasail_probe
        lib = parasail_probe.load()
        if lib is None:
            return None
        from concurrent.futures import ThreadPoolExecutor
        sample = 65536
        qs = [qbuf[k * LEN:(k + 1) * LEN].tobytes() for k in range(sample)]
        rs = [rbuf[k * LEN:(k + 1) * LEN].tobytes() for k in range(sample)]
        per = (sample + cores - 1) // cores
        def work(c):
            return parasail_probe.align_batch(lib, b"sw_striped_16", qs[c * per:(c + 1) * per], rs[c * per:(c + 1) * per],
                                              OPEN, EXT, b"ACGT", MATCH, MISMATCH)
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, range(cores)))                # warm-up
            t0 = time.perf_counter()
            parts = list(ex.map(work, range(cores)))
            t = time.perf_counter() - t0
        out = np.array([x for p in parts for x in p], dtype=np.int32)
        return {"value": round(sample * LEN * LEN / t / 1e9, 3), "unit": "GCUPS", "cores": int(cores), "kind": "reference",
                "sample": "%d of the same 150x150 pairs, system libparasail parasail_sw_striped_16 via ctypes, one thread per core, "
                          "%.2f s wall" % (sample, t)}, out
    except Exception:
        return None


def cpu_baseline(qbuf, qoff, rbuf, roff):
    """The CPU port of the reference's kernel class (Farrar striped int16, AVX2 + OpenMP), all host
    cores, on a bounded sample of the same workload (or the real library, if the box has one)."""
    ref = cpu_reference_baseline(qbuf, rbuf, usable_cores())
    if ref is not None:
        return ref
    from oracle import oracle as orc
    m = orc.Matrix.create("ACGT", MATCH, MISMATCH)
    cores = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))

    def run(npairs):
        t0 = time.perf_counter()
        out, used = orc.cpu_sw_striped16_batch(qbuf[: npairs * LEN], qoff[: npairs + 1], rbuf[: npairs * LEN],
                                               roff[: npairs + 1], OPEN, EXT, m, threads=cores)
        return time.perf_counter() - t0, used, out
    run(2048)                                           # warm-up (thread pool, page-in)
    t, used, _ = run(16384)
    rate = 16384 / max(t, 1e-6)
    sample = int(min(N_PAIRS, max(16384, rate * 2.0)))  # ~2 s wall
    t, used, out = run(sample)
    cpu = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(sample * LEN * LEN / t / 1e9, 3), "unit": "GCUPS", "cores": int(used), "kind": "port",
            "sample": "%d of the same 150x150 pairs, restated CPU baseline (not parasail): striped int16 AVX2 + OpenMP "
                      "(oracle/pmx_striped_cpu.c, gcc -O2 -march=x86-64-v3 -fopenmp), %s, %.2f s wall" % (sample, cpu, t)}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=N_PAIRS, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_pkg()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        print("bench.py: --gpus %d needs torch.distributed.run with %d processes" % (args.gpus, args.gpus),
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # PMX_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share
    # devices, records are gathered through host memory); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("PMX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # PMX_BENCH_FORCE_DIST=1 (rehearsal on a one-GPU box): take the exchange path with a single rank
    multi = world > 1 or bool(os.environ.get("PMX_BENCH_FORCE_DIST"))
    if multi:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    n = args.pairs
    qbuf, qoff, rbuf, roff = make_cfg2_inputs(n, SEED + rank)     # each rank its own batch (weak scaling)
    d_q = torch.from_numpy(qbuf).to(dev)
    d_r = torch.from_numpy(rbuf).to(dev)
    d_qo = torch.from_numpy(qoff).to(dev)
    d_ro = torch.from_numpy(roff).to(dev)
    d_out = [torch.zeros((n, 4), dtype=torch.int32, device=dev) for _ in range(2)]
    matrix = pkg.Matrix.create(b"ACGT", MATCH, MISMATCH)
    cfg = pkg.pmx_config_t(pkg.MODE_SW, 0, OPEN, EXT, 16, 0, matrix.inner)

    from importlib import import_module
    sharding = import_module("parasail_rs_amd.sharding")
    counts = [n] * world
    comm_stream = torch.cuda.Stream(device=dev) if multi else None
    pending = []

    def step(k, events=None):
        out = d_out[k % 2]
        stream = torch.cuda.current_stream(dev)
        while len(pending) >= 2:          # the gather that last read this output buffer must be done
            fin, work = pending.pop(0)
            work.wait()
        if events is not None:
            events[0].record(stream)
        pkg.align_batch_device(cfg, n, d_q.data_ptr(), d_qo.data_ptr(), d_r.data_ptr(), d_ro.data_ptr(),
                               LEN, LEN, out.data_ptr(), None, stream.cuda_stream)
        if events is not None:
            events[1].record(stream)
        if multi:
            # exchange step: records of this step go to rank 0 while the next step computes
            if backend == "nccl":
                done = torch.cuda.Event()
                done.record(stream)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(done)
                    pending.append(sharding.gather_records(out, counts, dst=0, async_op=True))
            else:
                pending.append(sharding.gather_records(out.cpu(), counts, dst=0, async_op=True))

    def drain():
        while pending:
            fin, work = pending.pop(0)
            work.wait()
        if comm_stream is not None:
            torch.cuda.current_stream(dev).wait_stream(comm_stream)

    for k in range(args.warmup):
        step(k)
    drain()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, evs[k])
    drain()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if evs else float("nan")
    kernel = pkg.lib.pmx_last_kernel().decode()        # name + shape + arithmetic variant of what this thread just launched

    cells_per_rank_step = n * LEN * LEN
    total_cells = cells_per_rank_step * world * args.steps
    gcups = total_cells / elapsed / 1e9

    if rank == 0:
        achieved = ALGO_BYTES_PER_PAIR * n / (kern_ms * 1e-3) / 1e9
        valu_peak_gcups = SIMD_CYCLES_PER_S / CYCLES_PER_128_CELLS * 128.0 / 1e9
        kern_gcups = cells_per_rank_step / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "GCUPS (cell updates/s) local-affine SW, 1M 150x150 pairs, 1/2/4/8 GPUs",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "cfg2: %d pairs/GPU x (150 bp x 150 bp) i.i.d. DNA, sw_striped_16 (score + end "
                                   "positions), Matrix::create(ACGT,2,-3), gaps 5/2" % n,
                       "pairs_per_gpu": n, "kernel": kernel, "inputs": "resident in HBM",
                       "exchange": "none" if world == 1 else "%s gather of 16-B records to rank 0, overlapped" %
                                   ("RCCL" if backend == "nccl" else backend)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic_bytes() if n == N_PAIRS else None,
                         "kernel_ms": round(kern_ms, 4),
                         "note": "312 algorithmic B/pair; the path is VALU-bound, see roofline_valu"},
            "roofline_valu": {"bound": "valu", "achieved": round(kern_gcups, 2), "peak": round(valu_peak_gcups, 1),
                              "unit": "GCUPS", "frac": round(kern_gcups / valu_peak_gcups, 4),
                              "model": "per 128 cells: 5.5 VOP3/VOP3P x 4 cycles + 3 VOP2 x 2 cycles = 28 SIMD cycles; "
                                       "1024 SIMDs x 2.4 GHz (measured issue rates, profiles/r01)"},
        }
        if world == 1 and not multi and n == N_PAIRS and not args.no_cpu_baseline:
            # the same batch handed over in host memory (H2D + kernels + D2H inside): reported beside, never as `value`
            al = pkg.Aligner.new().local().matrix(matrix).gap_open(OPEN).gap_extend(EXT).solution_width(16).build()
            al.align_batch_packed(qbuf, qoff, rbuf, roff)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); al.align_batch_packed(qbuf, qoff, rbuf, roff); ts.append(time.perf_counter() - t0)
            line["pcie_inclusive"] = {"value": round(cells_per_rank_step / min(ts) / 1e9, 1), "unit": "GCUPS",
                                      "ms": round(min(ts) * 1e3, 3), "entry": "pmx_align_batch (pageable host buffers)"}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_out = cpu_baseline(qbuf, qoff, rbuf, roff)
            line["cpu_baseline"] = cb
            got = d_out[(args.steps - 1) % 2][: len(cpu_out), :3].cpu().numpy()
            line["cpu_baseline"]["agrees_with_gpu"] = bool((got == cpu_out).all())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
