#!/usr/bin/env python3
"""Host time per call of the archive loop (reset, scan, resolve, collect of the
capture before) on captures of a given size, two contexts, two-stream schedule:
what a recording costs the host whatever its length.  Diagnostic.

usage: host_overhead.py [samples]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sora_amd import AdsbContext, synth                      # noqa: E402
from sora_amd._lib import ENC_UC8                            # noqa: E402
from sora_amd.sharded import LocalDecoder                    # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 120000000
dev = torch.device("cuda", 0)
p = synth.default_params(5, n_samples=n)
p.encoding = ENC_UC8
fr = synth.make_frames(p)
d = torch.empty(2 * n + 16, dtype=torch.uint8, device=dev)
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, fr)
torch.cuda.synchronize()
mf = n // 1024 + 65536
ss = torch.cuda.Stream(device=dev)
decs = []
for _ in range(2):
    rs = torch.cuda.Stream(device=dev, priority=-1)
    c = AdsbContext(n, device=0, max_squitters=mf, max_frames=mf, stream=ss.cuda_stream,
                    wire_frames=True, resolve_stream=rs.cuda_stream)
    decs.append(LocalDecoder(c, mf))
T = {}


def tm(name, f, *a, **k):
    t = time.perf_counter()
    r = f(*a, **k)
    T[name] = T.get(name, 0) + time.perf_counter() - t
    return r


K, W = 2000, 200
prev = None
for it in range(K + W):
    if it == W:
        T.clear()
        t0 = time.perf_counter()
    dec = decs[it & 1]
    c = dec.ctx
    tm("reset", c.reset)
    tm("scan", c.scan, d, 2 * n, ENC_UC8, base_position=0, n_owned=n)
    tm("resolve", c.resolve)
    if prev is not None:
        tm("collect", prev.finish)
    prev = dec
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n} samples: {dt / K * 1e6:.1f} us per capture;",
      {k: round(v / K * 1e6, 1) for k, v in T.items()}, "frames", prev.n)
