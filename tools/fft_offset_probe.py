#!/usr/bin/env python3
"""tools/fft_offset_probe.py — how the speed of the Z + Y and X passes depends on WHERE in memory the 768^3 mesh starts: one large
torch allocation, the mesh placed at a series of byte offsets inside it (shq_pm_slab2_fft_yz / _xgreen take any device pointer)."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq
from shenqi_amd import capi
N = 768
zp = capi.hip.shq_fft3d_pitch(N) if hasattr(capi.hip, "shq_fft3d_pitch") else 776
words = N * N * zp
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = sq.Context(0, stream=stream.cuda_stream)
span = 1 << 30
big = torch.zeros(words + span // 8 + 16, dtype=torch.float64, device=dev)
base = big.data_ptr()
print("allocation base 0x%x (mod 2^30 = 0x%x)" % (base, base & (span - 1)))
pmp = sq.PMParams(N, 0, 1.0, 1.5, 43.0071)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(n):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n
offs = [0, 1 << 12, 1 << 16, 1 << 20, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29, (1 << 29) + (1 << 21)]
for off in offs:
    p = base + off
    tyz = timed(lambda: capi.check(capi.hip.shq_pm_slab2_fft_yz(ctx.h, N, C.c_void_p(p), N, 0)))
    tx = timed(lambda: capi.check(capi.hip.shq_pm_slab2_xgreen(ctx.h, C.byref(pmp), C.c_void_p(p), 0, N)))
    big.zero_()
    print("offset 0x%09x: Z+Y forward %.3f ms, X fused %.3f ms" % (off, tyz, tx), flush=True)
