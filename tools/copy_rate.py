#!/usr/bin/env python3
"""tools/copy_rate.py — what a plain read + write stream reaches on this card (the yardstick for the FFT passes, which read and write
the 3.66 GB mesh once each): torch copies of 1 / 3.66 GB, contiguous and in 64-byte pieces 6208 bytes apart (the Y pass's shape)."""
import torch
dev = "cuda:0"
def rate(fn, nbytes, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return nbytes * n / (a.elapsed_time(b) * 1e-3) / 1e12
for gb in (1.0, 3.66):
    n = int(gb * 1e9 / 8)
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_()
    y = torch.empty_like(x)
    print("contiguous copy %.2f GB: %.2f TB/s (read + write bytes)" % (gb, rate(lambda: y.copy_(x), 2 * n * 8)))
    print("fill %.2f GB: %.2f TB/s" % (gb, rate(lambda: y.zero_(), n * 8)))
# strided: [768][776*... ] view: rows of 776 complex; copy a [rows, 4-complex] column block = 64-byte pieces
N, zpc = 768, 388
m = torch.empty((N * N, zpc, 2), dtype=torch.float64, device=dev).normal_()
o = torch.empty_like(m)
def strided():
    # transpose-like traffic: every 4-column block separately would be too many launches; one permuted copy moves the same 64-byte pieces
    o.view(N, N, zpc // 4, 4, 2).copy_(m.view(N, N, zpc // 4, 4, 2))
print("blocked copy (control) %.2f TB/s" % rate(strided, 2 * m.numel() * 8))
