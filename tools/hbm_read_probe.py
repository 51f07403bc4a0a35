#!/usr/bin/env python3
"""Diagnostic: what a plain streaming read of 14.4 GB reaches on this GPU
(torch reductions over int32 / float32 / int64 views), to put the scan
kernel's 4.3-4.4 TB/s beside the practical HBM ceiling rather than the 8 TB/s
of the data sheet."""
import time
import torch

n = 14_400_000_000
x = torch.empty(n, dtype=torch.uint8, device="cuda")
x.random_(0, 255)
for name, view in (("int32 sum", x.view(torch.int32)), ("float32 max", x.view(torch.float32)),
                   ("int64 sum", x.view(torch.int64)), ("int32 max", x.view(torch.int32))):
    op = (lambda v: v.sum()) if "sum" in name else (lambda v: v.max())
    for _ in range(3):
        op(view)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); op(view); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{name:12s} min {min(ts):.3f} ms = {n / min(ts) / 1e6:.0f} GB/s, mean {sum(ts) / len(ts):.3f} ms")
y = torch.empty(n // 2, dtype=torch.uint8, device="cuda")
for _ in range(3):
    y.copy_(x[:n // 2])
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); y.copy_(x[:n // 2]); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(f"copy 7.2 GB  min {min(ts):.3f} ms = {n / min(ts) / 1e6:.0f} GB/s read+write")
