#!/usr/bin/env python3
"""Diagnostic: resolve-stage time (first-seen, filter, flatten, finalize) of a
variant library on the configs[1] capture.
    python tools/resolve_bench.py LIBNAME [samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sora_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "sora_amd", "lib", sys.argv[1])
from sora_amd import ENC_UC8, AdsbContext, synth
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 7_200_000_000
p = synth.default_params(2, n_samples=n)
fr = synth.make_frames(p)
d = torch.empty(2 * n + 16, dtype=torch.uint8, device="cuda")
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, fr)
ctx = AdsbContext(n, max_squitters=n // 1024 + 65536)
ms = []
for _ in range(8):
    ctx.reset(); ctx.scan(d, 2 * n, ENC_UC8); ctx.synchronize()
    t = time.perf_counter(); ctx.resolve(); ctx.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
print(sys.argv[1], "resolve ms", [round(x, 3) for x in ms[2:]], "scan", round(ctx.last_scan_ms(), 3), "frames", ctx.frame_count())
