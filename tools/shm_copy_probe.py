#!/usr/bin/env python3
"""Diagnostic: how long the CPU copy of one rank's frame list (pinned staging
buffer -> the node's shared /dev/shm segment) takes with 1 and with 8 ranks
copying at once (sora_amd/sharded.py, the path taken when the segment cannot be
page-locked).    python tools/shm_copy_probe.py [MB] [procs...]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import torch


def worker(rank, mb, n_iter, q, barrier, nt):
    if nt:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from sora_amd import _lib
        copy = _lib.lib().sora_adsb_host_copy
    n = mb << 20
    path = f"/dev/shm/sora_probe_{os.getppid()}_{rank}"
    with open(path, "wb") as fh:
        fh.truncate(n)
    mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(n,))
    os.unlink(path)
    dst = torch.from_numpy(mm)
    src = torch.empty(n, dtype=torch.uint8)
    src.fill_(rank + 1)
    dst.copy_(src)                      # touch the pages
    barrier.wait()
    t = []
    for _ in range(n_iter):
        t0 = time.perf_counter()
        for a in range(0, n, 16 << 20):
            if nt:
                copy(dst.data_ptr() + a, src.data_ptr() + a, min(16 << 20, n - a))
            else:
                dst[a:a + (16 << 20)].copy_(src[a:a + (16 << 20)])
        t.append((time.perf_counter() - t0) * 1e3)
    q.put((rank, min(t), sum(t) / len(t)))


if __name__ == "__main__":
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 83
    torch.set_num_threads(1)
    for procs, nt in [(int(x), nt) for x in (sys.argv[2:] or ["1", "8"]) for nt in (0, 1)]:
        q, barrier = mp.Queue(), mp.Barrier(procs)
        ps = [mp.Process(target=worker, args=(r, mb, 20, q, barrier, nt)) for r in range(procs)]
        for p in ps:
            p.start()
        res = sorted(q.get() for _ in ps)
        for p in ps:
            p.join()
        print(f"{'non-temporal' if nt else 'memcpy      '} {procs} procs x {mb} MB: min {min(r[1] for r in res):.2f} ms, "
              f"mean {sum(r[2] for r in res) / procs:.2f} ms, worst mean {max(r[2] for r in res):.2f} ms")
