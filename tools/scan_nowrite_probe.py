#!/usr/bin/env python3
"""Diagnostic: the scan kernel's time with and without its record stores
(a record capacity of one per region drops nearly all of them; the counts
still come out, the resolve is not run).  Tells whether the 1.3 GB of record
writes cost time next to the 14.4 GB of reads.
    python tools/scan_nowrite_probe.py [samples]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sora_amd import ENC_UC8, AdsbContext, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 7_200_000_000
p = synth.default_params(2, n_samples=n)
fr = synth.make_frames(p)
d = torch.empty(2 * n + 16, dtype=torch.uint8, device="cuda")
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, fr)
for label, kw in (("with records", {}), ("records dropped", {"max_candidates": 40000}),
                  ("with records", {}), ("records dropped", {"max_candidates": 40000})):
    ctx = AdsbContext(n, max_squitters=n // 1024 + 65536, **kw)
    ms = []
    for _ in range(8):
        ctx.reset(); ctx.scan(d, 2 * n, ENC_UC8); ctx.synchronize(); ms.append(ctx.last_scan_ms())
    print(label, "scan ms", [round(x, 3) for x in ms[2:]], "GB/s", round(2 * n / min(ms) / 1e6, 1))
    ctx.close()
