#!/usr/bin/env python3
"""tools/fft_vmm_probe.py — does the way the mesh is allocated change how fast the FFT passes stream it?  Four hipMalloc'ed meshes and
four meshes mapped through the virtual-memory API (hipMemCreate at the recommended granularity), Z + Y forward and fused X timed on each."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shenqi_amd as sq
from shenqi_amd import capi
hip = C.CDLL("libamdhip64.so")
N = 768
zp = 776
nbytes = N * N * zp * 8
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = sq.Context(0, stream=stream.cuda_stream)
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

def probe(p, tag):
    hip.hipMemsetAsync(C.c_void_p(p), 0, C.c_size_t(nbytes), C.c_void_p(stream.cuda_stream))
    tyz = timed(lambda: capi.check(capi.hip.shq_pm_slab2_fft_yz(ctx.h, N, C.c_void_p(p), N, 0)))
    hip.hipMemsetAsync(C.c_void_p(p), 0, C.c_size_t(nbytes), C.c_void_p(stream.cuda_stream))
    tx = timed(lambda: capi.check(capi.hip.shq_pm_slab2_xgreen(ctx.h, C.byref(pmp), C.c_void_p(p), 0, N)))
    print("%-10s 0x%x: Z+Y forward %.3f ms, X fused %.3f ms" % (tag, p, tyz, tx), flush=True)

keep = []
for i in range(4):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    keep.append(p)
    probe(p.value, "hipMalloc")

class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", C.c_ubyte * 8)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]
prop = Prop()
prop.type = 1          # hipMemAllocationTypePinned
prop.location.type = 1 # hipMemLocationTypeDevice
prop.location.id = 0
gran = C.c_size_t()
rc = hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1)  # recommended
print("granularity rc", rc, gran.value)
g = max(gran.value, 1 << 21)
size = (nbytes + g - 1) // g * g
for i in range(4):
    va = C.c_void_p()
    rc = hip.hipMemAddressReserve(C.byref(va), C.c_size_t(size), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0))
    h = C.c_void_p()
    rc2 = hip.hipMemCreate(C.byref(h), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0))
    rc3 = hip.hipMemMap(va, C.c_size_t(size), C.c_size_t(0), h, C.c_ulonglong(0))
    acc = Access()
    acc.location.type, acc.location.id, acc.flags = 1, 0, 3
    rc4 = hip.hipMemSetAccess(va, C.c_size_t(size), C.byref(acc), C.c_size_t(1))
    if rc or rc2 or rc3 or rc4:
        print("vmm failed", rc, rc2, rc3, rc4)
        break
    probe(va.value, "vmm")
