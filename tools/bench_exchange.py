"""Throughput of the exchange loops on one task's arrays: n particles (160-byte records; a fifth of them gas with 176-byte slots), every
particle's target drawn among ntask tasks, so (ntask - 1) / ntask of them leave.  Times shq_exchange_plan, shq_exchange_pack and the
shq_slots_gc that follows, and prices the pack against HBM: bytes read + written per leaving particle."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq  # noqa: E402
from shenqi_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ntask = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(1)
f = capi.PARTICLE_DTYPE.fields
L = capi.ExchangeLayout()
L.part_elsize, L.off_flags, L.off_type, L.off_pi = capi.PARTICLE_DTYPE.itemsize, f["Flags"][1], f["Type"][1], f["PI"][1]
L.slot_elsize[0] = capi.SPH_DTYPE.itemsize
L.off_reverselink = 0
P = np.zeros(n, dtype=capi.PARTICLE_DTYPE)
P["Type"] = np.where(rng.random(n) < 0.2, 0, 1)
gas = np.flatnonzero(P["Type"] == 0)
P["PI"][gas] = np.arange(len(gas))
P["ID"] = np.arange(n) + 1
S = np.zeros(len(gas), dtype=capi.SPH_DTYPE)
S["ReverseLink"] = gas
dev = "cuda:0"
c = sq.Context(0)
target = torch.from_numpy(rng.integers(0, ntask, n).astype(np.int32)).to(dev)
esz, ssz = int(L.part_elsize), int(L.slot_elsize[0])
for rep in range(3):
    d_parts = torch.from_numpy(P.view(np.uint8).reshape(-1)).to(dev)
    d_slot = torch.from_numpy(S.view(np.uint8).reshape(-1)).to(dev)
    sp = (C.c_void_p * 6)(d_slot.data_ptr(), None, None, None, None, None)
    tg = (capi.ExchangeEntry * ntask)()
    nex, last = C.c_int64(), C.c_int64()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(capi.hip.shq_exchange_plan(c.h, C.byref(L), d_parts.data_ptr(), n, target.data_ptr(), 0, ntask, 0, C.byref(nex), C.byref(last), tg))
    c.synchronize()
    t1 = time.perf_counter()
    togo = np.array([[tg[t].base] + list(tg[t].slots) for t in range(ntask)], dtype=np.int64)
    off = np.zeros_like(togo)
    off[1:] = np.cumsum(togo[:-1], axis=0)
    e = (capi.ExchangeEntry * ntask)()
    for k in range(ntask):
        e[k].base = int(off[k, 0])
        for t in range(6):
            e[k].slots[t] = int(off[k, 1 + t])
    nb, ns = int(togo[:, 0].sum()), int(togo[:, 1].sum())
    pb = torch.empty(max(nb, 1) * esz, dtype=torch.uint8, device=dev)
    sb = torch.empty(max(ns, 1) * ssz, dtype=torch.uint8, device=dev)
    bp = (C.c_void_p * 6)(sb.data_ptr(), None, None, None, None, None)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    capi.check(capi.hip.shq_exchange_pack(c.h, C.byref(L), d_parts.data_ptr(), sp, n, e, ntask, pb.data_ptr(), bp))
    c.synchronize()
    t3 = time.perf_counter()
    nn = C.c_int64(n)
    sz = (C.c_int64 * 6)(len(gas), 0, 0, 0, 0, 0)
    compact = (C.c_int * 6)(1, 0, 0, 0, 0, 0)
    capi.check(capi.hip.shq_slots_gc(c.h, C.byref(L), d_parts.data_ptr(), C.byref(nn), n, sp, sz, compact))
    c.synchronize()
    t4 = time.perf_counter()
    moved = 2.0 * (nb * esz + ns * ssz)
    kept = 2.0 * (nn.value * esz + sz[0] * ssz)
    print("n = %d, %d tasks: %d leave (%d gas); plan %.2f ms, pack %.2f ms = %.0f GB/s of record traffic, slots_gc %.2f ms = %.0f GB/s (%d stay)" %
          (n, ntask, nb, ns, (t1 - t0) * 1e3, (t3 - t2) * 1e3, moved / (t3 - t2) / 1e9, (t4 - t3) * 1e3, kept / (t4 - t3) / 1e9, nn.value), flush=True)
c.close()
