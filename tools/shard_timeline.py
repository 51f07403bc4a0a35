#!/usr/bin/env python3
"""Where a short shard's step goes on the N>1 code path, from HIP events.

bench.py --samples 9e8 --force-collective with its two-stream schedule, with an
event recorded on the stream behind every operation of a step (rocprofv3
serialises the queues, so its timeline does not show what overlaps what; the
events do).  Prints, per step, the times in microseconds since the step's scan
began.  Diagnostic only.

usage: shard_timeline.py [samples] [local|collective] [steps] [warmup steps]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sora_amd                                              # noqa: E402
from sora_amd import AdsbContext, synth                      # noqa: E402
from sora_amd._lib import ENC_UC8                            # noqa: E402
from sora_amd.sharded import LocalDecoder, ShardedDecoder    # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 900000000
    mode = sys.argv[2] if len(sys.argv) > 2 else "collective"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if mode == "collective":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    params = synth.default_params(2, n_samples=n)
    frames = synth.make_frames(params)
    d_iq = torch.empty(2 * n + 16, dtype=torch.uint8, device=dev)
    synth.fill_device(d_iq.data_ptr(), n, ENC_UC8, params.seed, frames, first=0)
    torch.cuda.synchronize()
    max_frames = n // 1024 + 65536
    scan_stream = torch.cuda.Stream(device=dev, priority=0)
    decs, rstreams = [], []
    for _ in range(2):
        rs = torch.cuda.Stream(device=dev, priority=-1)
        ctx = AdsbContext(n, device=0, max_squitters=max_frames, max_frames=max_frames,
                          stream=scan_stream.cuda_stream, wire_frames=True,
                          resolve_stream=rs.cuda_stream)
        rstreams.append(rs)
        decs.append(ShardedDecoder(ctx, 0, 1, dev, max_frames, stream=rs)
                    if mode == "collective" else LocalDecoder(ctx, max_frames))
    job = (d_iq, 2 * n, ENC_UC8, 0, n)

    def ev(stream):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    log = []

    def exchange(d, rs, marks):
        if mode != "collective":
            d.ctx.resolve()
            marks.append(("resolve", ev(rs)))
            return
        with torch.cuda.stream(rs):
            d.ctx.export_block(d.d_blk, d.block)
            marks.append(("export", ev(rs)))
            dist.all_gather_into_tensor(d.d_blk_all, d.d_blk)
            marks.append(("gather", ev(rs)))
            d.ctx.add_known_blocks(d.d_blk_all, 1, d.block, skip_block=0)
            marks.append(("add", ev(rs)))
            d.ctx.resolve()
            marks.append(("resolve", ev(rs)))
            d.h_exported.copy_(d.d_blk_all.view(1, d.block)[:, 0], non_blocking=True)
            marks.append(("copy8", ev(rs)))

    import time
    base = ev(scan_stream)
    prev = None
    for i in range(steps + warm):
        d, rs = decs[i & 1], rstreams[i & 1]
        marks = []
        t0 = time.perf_counter()
        marks.append(("s0", ev(scan_stream)))
        d.begin(*job)
        marks.append(("s1", ev(scan_stream)))
        t1 = time.perf_counter()
        exchange(d, rs, marks)
        t2 = time.perf_counter()
        if prev is not None:
            prev[0].finish()
            prev[2].append(("done", ev(prev[1])))
        t3 = time.perf_counter()
        cur = (d, rs, marks, (t0, t1, t2, t3))
        log.append(cur)
        prev = cur
    prev[0].finish()
    prev[2].append(("done", ev(prev[1])))
    torch.cuda.synchronize()
    print(f"{mode}, {n} samples; columns: microseconds since the first event")
    t_first = log[warm][3][0]
    for i, (d, rs, marks, host) in enumerate(log[warm:warm + steps]):
        row = " ".join(f"{name}={base.elapsed_time(e) * 1e3:8.0f}" for name, e in marks)
        h = " ".join(f"{(t - t_first) * 1e6:7.0f}" for t in host)
        print(f"step {i:2d} gpu: {row}")
        print(f"        host(begin, exchange, finish-prev, end): {h}")
    s1 = [base.elapsed_time(m[2][1][1]) for m in log[warm:warm + steps]]
    print("scan ends apart (us):", " ".join(f"{(b - a) * 1e3:.0f}" for a, b in zip(s1, s1[1:])))
    for d in decs:
        d.close() if hasattr(d, "close") else None
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
