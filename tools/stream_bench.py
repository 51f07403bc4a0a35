#!/usr/bin/env python3
"""PCIe-inclusive rate of the streaming ingest (sora_adsb_stream_*): a capture
that lives in HOST memory is pulled through pinned double buffers, copied to
the device and decoded block by block.  Not bench.py's `value` (that one has
its input resident in HBM); DESIGN.md quotes this number beside it.

    python tools/stream_bench.py [samples] [block_samples]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                           # noqa: E402
import torch                                                 # noqa: E402
from sora_amd import ENC_UC8, AdsbContext, _lib, synth       # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000_000
block = int(float(sys.argv[2])) if len(sys.argv) > 2 else 64 << 20

p = synth.default_params(2, n_samples=n)
frames = synth.make_frames(p)
d = torch.empty(2 * n, dtype=torch.uint8, device="cuda")
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, frames)
host = d.cpu().numpy()                                       # pageable host memory
del d
base = host.ctypes.data
at = [0]


def read(_user, dst, nbytes):                                # memcpy = a page-cache fread
    k = min(nbytes, host.size - at[0])
    if k > 0:
        C.memmove(dst, base + at[0], k)
    at[0] += k
    return k


L = _lib.lib()
cb = _lib.READ_FN(read)
with AdsbContext(block + 239, max_squitters=block // 512 + 65536) as ctx:
    best = None
    for rep in range(3):
        at[0] = 0
        ctx.reset()
        total = 0
        t0 = time.perf_counter()
        rc = L.sora_adsb_stream_begin(ctx._h, ENC_UC8, block, cb, None)
        assert rc == 0, rc
        while True:
            ptr, k = C.c_void_p(), C.c_size_t()
            rc = L.sora_adsb_stream_next(ctx._h, C.byref(ptr), C.byref(k))
            if rc == _lib.END:
                break
            assert rc == 0, (rc, L.sora_adsb_last_error(ctx._h))
            total += k.value
        L.sora_adsb_stream_end(ctx._h)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
print(json.dumps({"what": "streaming ingest from host memory (PCIe inclusive, pageable "
                          "source memcpy'd into pinned blocks by one thread)",
                  "samples": n, "block_samples": block, "frames": total,
                  "seconds": round(best, 4),
                  "Msamples_per_s": round(n / best / 1e6, 1),
                  "GB_per_s": round(2 * n / best / 1e9, 2)}))
