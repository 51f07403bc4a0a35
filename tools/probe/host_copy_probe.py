#!/usr/bin/env python3
"""D2H copy rate of a frame list (6 MB) into different kinds of host memory:
torch pinned (hipHostMalloc), a /dev/shm mapping registered with
hipHostRegister (what ShardedDecoder's shared segment is), an anonymous
mapping registered the same way, with and without MADV_HUGEPAGE.  Diagnostic."""
import mmap
import os
import sys
import time

import numpy as np
import torch

NB = int(float(sys.argv[1])) if len(sys.argv) > 1 else 6 << 20
SEG = 64 << 20
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sora_amd import _lib                                    # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
src = torch.randint(0, 255, (NB,), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()


def rate(dst, label):
    ts = []
    with torch.cuda.stream(st):
        for _ in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record()
            dst[:NB].copy_(src, non_blocking=True)
            b.record()
            b.synchronize()
            ts.append((a.elapsed_time(b) * 1e3, (time.perf_counter() - t0) * 1e6))
    ev = np.median([t[0] for t in ts[5:]])
    wall = np.median([t[1] for t in ts[5:]])
    ok = bool((dst[:NB] == src.cpu()).all())
    print(f"{label:44s} {ev:7.0f} us on the stream, {wall:7.0f} us on the host, "
          f"{NB / ev / 1e3:5.1f} GB/s, pinned={dst.is_pinned()} equal={ok}")


rate(torch.empty(SEG, dtype=torch.uint8, pin_memory=True), "torch pinned")

def registered(buf, label, flags=0):
    arr = np.frombuffer(buf, dtype=np.uint8)
    arr[:] = 0
    rc = L.sora_adsb_host_register(0, arr.ctypes.data, arr.nbytes)
    if rc != _lib.OK:
        print(label, "sora_adsb_host_register:", rc)
        return
    rate(torch.from_numpy(arr), label)
    L.sora_adsb_host_unregister(0, arr.ctypes.data)

name = f"/dev/shm/sora_probe_{os.getpid()}"
with open(name, "wb") as fh:
    fh.truncate(SEG)
fd = os.open(name, os.O_RDWR)
os.unlink(name)
registered(mmap.mmap(fd, SEG), "/dev/shm mapping, registered")
registered(mmap.mmap(-1, SEG), "anonymous mapping, registered")
m = mmap.mmap(-1, SEG + (2 << 20))
if hasattr(mmap, "MADV_HUGEPAGE"):
    m.madvise(mmap.MADV_HUGEPAGE)
registered(m, "anonymous mapping, MADV_HUGEPAGE, registered")
for f in ("enabled", "shmem_enabled"):
    try:
        print(f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except OSError as e:
        print(f, e)
