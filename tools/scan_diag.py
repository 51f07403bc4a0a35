#!/usr/bin/env python3
"""Diagnostics for the scan kernel (not part of the product or the tests).

    python tools/scan_diag.py stamps [samples]   phase shares from the
                                                  -DSORA_SCAN_STAMPS build
    python tools/scan_diag.py run [samples] [n]  n plain scans (for rocprofv3 --pmc)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

mode = sys.argv[1] if len(sys.argv) > 1 else "run"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_200_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3

import torch                                  # noqa: E402
from sora_amd import _lib                     # noqa: E402

if mode == "stamps":
    _lib.LIB_PATH = os.path.join(ROOT, "sora_amd", "lib", "libsora_adsb_gpu_stamps.so")
if mode == "x":                               # `make -C sora_amd/csrc x XFLAGS=...`
    _lib.LIB_PATH = os.path.join(ROOT, "sora_amd", "lib", os.environ.get(
        "SORA_X_LIB", "libsora_adsb_gpu_x.so"))
import sora_amd                               # noqa: E402
from sora_amd import ENC_UC8, AdsbContext, synth   # noqa: E402

p = synth.default_params(2, n_samples=n)
fr = synth.make_frames(p)
d = torch.empty(2 * n + 16, dtype=torch.uint8, device="cuda")
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, fr)
ctx = AdsbContext(n)
for _ in range(reps):
    ctx.reset()
    ctx.scan(d, 2 * n, ENC_UC8)
    ctx.synchronize()
ms = []
for _ in range(reps):
    ctx.reset()
    ctx.scan(d, 2 * n, ENC_UC8)
    ctx.synchronize()
    ms.append(ctx.last_scan_ms())
print("scan ms", min(ms), "GB/s", 2 * n / min(ms) / 1e6, os.environ.get("SORA_X_LIB", ""))
if mode == "stamps":
    L = _lib.lib()
    out = (C.c_uint64 * 8)()
    L.sora_adsb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    rc = L.sora_adsb_debug_stamps(ctx._h, out)
    names = (["load wait", "magnitudes + next loads", "compares", "predicate+scan",
              "barrier A", "queue", "barrier B", "candidates"]
             if os.environ.get("SORA_ADSB_SCAN_VARIANT", "1") == "1" else
             ["load wait", "magnitude+LDS store", "barrier 1", "LDS read+compare",
              "word store+barrier 2", "predicate+scan+queue", "candidates", "-"])
    tot = sum(out)
    for k, v in zip(names, out):
        print(f"{k:24s} {v:16d} {100.0 * v / max(tot, 1):6.2f} %")
ctx.close()
