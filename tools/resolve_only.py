#!/usr/bin/env python3
"""Diagnostic: the resolve alone (filter, offsets, flatten, finalize) on the
1-hour capture, for a library build.  python tools/resolve_only.py [LIB] [samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sora_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.LIB_PATH = os.path.join(ROOT, "sora_amd", "lib", sys.argv[1])
import torch, sora_amd
from sora_amd import ENC_UC8, AdsbContext, synth
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 7_200_000_000
p = synth.default_params(2, n_samples=n); fr = synth.make_frames(p)
d = torch.empty(2*n+16, dtype=torch.uint8, device="cuda")
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, fr)
mf = n // 1024 + 65536
ctx = AdsbContext(n, max_squitters=mf, max_frames=mf, wire_frames=True)
ctx.scan(d, 2*n, ENC_UC8); ctx.synchronize()
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.resolve(); ctx.synchronize()
    ts.append(time.perf_counter() - t0)
print(sys.argv[1:], "resolve ms min %.4f median %.4f" % (min(ts)*1e3, sorted(ts)[len(ts)//2]*1e3), "frames", ctx.frame_count())
