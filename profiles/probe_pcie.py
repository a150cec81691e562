"""Where the host-entry time of config 2 goes: raw PCIe rates of this box and pmx_align_batch / pmx_align_batch_2bit with
pageable, page-locked and pre-touched buffers.  Run on the GPU box: python profiles/probe_pcie.py"""
import ctypes as C, importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workloads as wl
pkg = importlib.import_module("parasail-rs_amd")
lib = pkg.lib

def best(f, k=5):
    f(); ts = []
    for _ in range(k):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)

res = {}
for mb in (16, 75, 300):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    def up(): d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    def down(): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    res["pinned_h2d_%dMB_GBps" % mb] = round((mb << 20) / best(up) / 1e9, 1)
    res["pinned_d2h_%dMB_GBps" % mb] = round((mb << 20) / best(down) / 1e9, 1)
    hp = torch.empty(mb << 20, dtype=torch.uint8); hp.fill_(1)
    def upp(): d.copy_(hp); torch.cuda.synchronize()
    res["pageable_h2d_%dMB_GBps" % mb] = round((mb << 20) / best(upp) / 1e9, 1)

c = wl.CFG2
n = c["n"]
qbuf, qoff, rbuf, roff = wl.make_cfg2(n)
al = pkg.Aligner.new().local().matrix(pkg.Matrix.create(c["matrix"][0].encode(), c["matrix"][1], c["matrix"][2])).gap_open(c["open"]).gap_extend(c["ext"]).solution_width(16).build()
cfg = al._config()
out = np.zeros(n, dtype=pkg.RECORD_DTYPE); out[:] = 0      # touched
cells = n * c["len"] ** 2
def call(): 
    rc = lib.pmx_align_batch(C.byref(cfg), n, qbuf.ctypes.data, qoff.ctypes.data, rbuf.ctypes.data, roff.ctypes.data, out.ctypes.data, None)
    assert rc == 0
t = best(call); res["bytes_pageable_ms"] = round(t * 1e3, 3)
ref = out.copy()
pkg.host_register(qbuf, qoff, rbuf, roff, out)
t = best(call); res["bytes_registered_ms"] = round(t * 1e3, 3)
pkg.host_unregister(qbuf, qoff, rbuf, roff, out)
q2, r2 = pkg.pack_2bit(qbuf), pkg.pack_2bit(rbuf)
def call2():
    rc = lib.pmx_align_batch_2bit(C.byref(cfg), n, q2.ctypes.data, qoff.ctypes.data, r2.ctypes.data, roff.ctypes.data, out.ctypes.data, None)
    assert rc == 0
t = best(call2); res["2bit_pageable_ms"] = round(t * 1e3, 3)
pkg.host_register(q2, qoff, r2, roff, out)
t = best(call2); res["2bit_registered_ms"] = round(t * 1e3, 3)
assert (out == ref).all()
pkg.host_unregister(q2, qoff, r2, roff, out)
for k in list(res):
    if k.endswith("_ms"): res[k.replace("_ms", "_GCUPS")] = round(cells / res[k] / 1e6, 1)
print(json.dumps(res, indent=1))
