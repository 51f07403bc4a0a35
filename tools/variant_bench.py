#!/usr/bin/env python3
"""Diagnostic: scan kernel time of a variant library.
    python tools/variant_bench.py LIBNAME [samples]"""
import os, sys
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
    ctx.reset(); ctx.scan(d, 2 * n, ENC_UC8); ctx.synchronize(); ms.append(ctx.last_scan_ms())
ctx.resolve(); k = len(ctx.collect())
print(sys.argv[1], "scan ms", [round(x, 3) for x in ms[2:]], "GB/s", round(2 * n / min(ms) / 1e6, 1), "frames", k)
