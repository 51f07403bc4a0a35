#!/usr/bin/env python3
"""bench.py -- Msamples/s of the ADS-B receive chain on N MI355X.

A step is one pass of the hot path over one synthetic capture that is already
resident in HBM: scan kernel -> (N>1: squitter-record exchange over RCCL) ->
first-seen merge, reply filter, re-slice of the accepted frames -> frame list
in host memory (N>1: every rank copies its frames over its own PCIe link into
a host segment the node's ranks share; rank 0 reads the whole list there).
Two captures are in flight (--inflight): the scan of one overlaps the resolve
and the host copy of the one before.  At N=1 the workload
is BASELINE.json configs[1]: a 1-hour 2 MS/s uc8 capture (7.2 G samples,
14.4 GB) with DF17/DF11 at mixed SNR.  For N>1 the default is configs[2]: the
SAME 1-hour capture in N haloed time shards (strong scaling: 900 M owned
positions + 239 halo samples per rank at N=8); --scaling weak gives every rank
a 1-hour shard of an N-hour capture instead.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec peak
# algorithmic traffic: every input byte once, 2 B per
# uc8/sc8 sample, 4 B per sc16 sample


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed steps are 0.7 s of GPU time: the two-deep pipeline's fill and
    # drain (one resolve and one scan that overlap nothing) are 0.5 % of that,
    # 3 % of 30 steps (3.63 ms per step at 30, 3.60 at 100 and more)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--samples", type=float, default=7.2e9,
                    help="samples of the capture (strong) or per GPU (weak); "
                         "default: 1 hour at 2 MS/s")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (BASELINE.json configs[2]): the one capture split over "
                         "the ranks; weak: one such capture per rank")
    ap.add_argument("--cpu-samples", type=float, default=120e6,
                    help="bounded sample for the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--encoding", choices=("uc8", "sc8", "sc16"), default="uc8",
                    help="uc8 is BASELINE.json's metric; sc8/sc16 are side measurements "
                         "(sc16 uses the dense configs[3] traffic)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="captures in flight (decoders on streams of their own); 1 = "
                         "strictly one after the other; 3 also moves the N>1 exchange "
                         "under the next scan (measured slower: its small kernels queue "
                         "behind the persistent scan kernel)")
    ap.add_argument("--gather", choices=("shm", "collective"), default="shm",
                    help="N>1: frame lists to rank 0 through a shared host segment "
                         "(every rank over its own PCIe link) or by an RCCL gather")
    ap.add_argument("--frames", choices=("wire", "full"), default="wire",
                    help="frame list as 28-byte wire frames (SORA_OPT_WIRE_FRAMES: the same "
                         "information, unpacked on the host on demand) or as 48-byte "
                         "struct sora_adsb_frame")
    ap.add_argument("--schedule", choices=("two-stream", "events"), default="two-stream",
                    help="two-stream: every capture's scan on ONE stream, back to back, and "
                         "everything behind a scan (exchange, resolve, frame copy) on a second "
                         "one (sora_adsb_set_resolve_stream); events: a stream per capture in "
                         "flight, ordered S(i) R(i) S(i+1) by events (rounds 1 and 2)")
    ap.add_argument("--resolve-priority", type=int, default=-1,
                    help="two-stream: priority of the second stream (torch's scale: -1 high, "
                         "0 as the scan's)")
    ap.add_argument("--overlap", action="store_true",
                    help="experiment: do not order the kernels of different captures "
                         "by events (a scan may share the GPU with a resolve)")
    ap.add_argument("--archive", type=int, default=0, metavar="R",
                    help="side mode, BASELINE.json configs[4]: R independent recordings per "
                         "GPU (120 M samples each by default, --archive-samples; uc8 and sc8 "
                         "alternating, a seed each) through one pair of contexts with "
                         "sora_adsb_reset() between them; a step is one pass over the R "
                         "recordings.  Replicas only: no collective at any N")
    ap.add_argument("--archive-samples", type=float, default=120e6)
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N>1 exchange code path even with one rank (rehearsal)")
    return ap.parse_args()


def cpu_all_cores(path, n_cpu):
    """SURVEY.md section 8(d)'s all-cores figure: one process of the compiled
    reference per core on disjoint chunks of the same window, each with a
    239-sample tail (the plane_iq_get_next driver with the reference's validator
    per chunk: the demodulator shards, the stateful rule does not, so this is a
    throughput figure, not a parity case)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_candidates_O2")
    if not os.path.exists(ref):
        return None
    cores = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "0")) or
                       len(os.sched_getaffinity(0))))
    per = -(-n_cpu // cores)
    procs = []
    t0 = time.perf_counter()
    for c in range(cores):
        first = c * per
        count = min(per + 239, n_cpu - first)
        cmd = (f"tail -c +{2 * first + 1} {path} | head -c {2 * count} | "
               f"{ref} --accepted-only | wc -l")
        procs.append(subprocess.Popen(["bash", "-c", cmd], stdout=subprocess.PIPE))
    frames = sum(int(p.communicate()[0]) for p in procs)
    dt = time.perf_counter() - t0
    return {"value": round(n_cpu / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores,
            "sample": f"the same window in {cores} chunks, one oracle/_ref/ref_candidates_O2 "
                      f"process each, {dt:.2f} s wall", "frames": frames}


def cpu_default_flags(path, n_cpu):
    """The same reference build with the flags its own configure uses
    (-std=c99 -g, no optimisation: reference configure:58, :242), on the first
    half of the window so that the leg stays bounded (about 10 s)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_sora_adsb")
    if not os.path.exists(ref):
        return None
    n = n_cpu // 2
    t0 = time.perf_counter()
    p = subprocess.run(["bash", "-c", f"head -c {2 * n} {path} | {ref} --adsb-decode "
                        "--adsb-from-raw --adsb-to-bitstring | wc -l"],
                       capture_output=True, check=True)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "sample": f"first {n} samples, oracle/_ref/ref_sora_adsb (reference sources, "
                      f"its own flags: gcc -std=c99 -g), {dt:.1f} s wall",
            "frames": int(p.stdout)}


def cpu_baseline(d_iq, n_cpu, gpu_accepted_in_window):
    """The reference itself (oracle/_ref, built from /root/reference by
    oracle/Makefile) timed on one host core over the first n_cpu samples of the
    same capture.  Falls back to the oracle port when oracle/_ref is absent."""
    raw = d_iq[:2 * n_cpu].cpu().numpy()
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_sora_adsb_O2")
    with tempfile.NamedTemporaryFile(suffix=".uc8", dir="/dev/shm" if
                                     os.path.isdir("/dev/shm") else None) as tf:
        raw.tofile(tf)
        tf.flush()
        if os.path.exists(ref):
            with open(tf.name, "rb") as fin:
                t0 = time.perf_counter()
                p = subprocess.run([ref, "--adsb-decode", "--adsb-from-raw",
                                    "--adsb-to-bitstring"], stdin=fin,
                                   capture_output=True, check=True)
                dt = time.perf_counter() - t0
            frames = p.stdout.count(b"\n")
            kind = "reference"
            what = "oracle/_ref/ref_sora_adsb_O2 (reference sources, gcc -O2), stdin"
            many = cpu_all_cores(tf.name, n_cpu)
            plain = cpu_default_flags(tf.name, n_cpu)
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob
            t0 = time.perf_counter()
            frames = len(ob.scan(raw.tobytes(), accepted_only=True, cap=1 << 22))
            dt = time.perf_counter() - t0
            kind = "port"
            what = "oracle/adsb_oracle.c (gcc -O2)"
            many = plain = None
    return {
        "value": round(n_cpu / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1,
        "kind": kind,
        "sample": f"first {n_cpu} samples ({n_cpu / 2e6:.0f} s) of the same capture, "
                  f"{what}, {dt:.1f} s wall",
        "frames": frames, "gpu_frames_same_window": gpu_accepted_in_window,
        "all_cores": many, "ref_default_flags": plain,
    }


def run_archive(args, json_fd):
    """configs[4]: archive reprocessing.  64 recordings over 8 GPUs are 8 per GPU;
    every rank works through its own R recordings (independent plane sets:
    sora_adsb_reset() between recordings, one capture = one recording) with two
    contexts in flight.  No data-path collective: replicas only."""
    import torch
    import torch.distributed as dist

    import sora_amd
    from sora_amd import AdsbContext, synth
    from sora_amd._lib import ENC_SC8, ENC_UC8
    from sora_amd.sharded import LocalDecoder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n, R = int(args.archive_samples), args.archive
    recs = []
    for i in range(R):
        enc = ENC_UC8 if i % 2 == 0 else ENC_SC8
        p = synth.default_params(5, n_samples=n, seed=0x5eed0000 + 97 * (rank * R + i))
        p.encoding = enc
        frames = synth.make_frames(p)
        d = torch.empty(2 * n + 16, dtype=torch.uint8, device=dev)
        synth.fill_device(d.data_ptr(), n, enc, p.seed, frames)
        recs.append((d, enc, len(frames)))
    torch.cuda.synchronize()
    max_frames = n // 1024 + 65536
    scan_stream = torch.cuda.Stream(device=dev, priority=0)
    ctxs, decs = [], []
    for _ in range(2):
        rs = torch.cuda.Stream(device=dev, priority=args.resolve_priority)
        ctx = AdsbContext(n, device=local, max_squitters=max_frames, max_frames=max_frames,
                          stream=scan_stream.cuda_stream, wire_frames=args.frames == "wire",
                          resolve_stream=rs.cuda_stream)
        ctxs.append(ctx)
        decs.append(LocalDecoder(ctx, max_frames))
    scan_ms, counts = [], [0] * R

    def one_pass():
        prev = None
        for i, (d, enc, _) in enumerate(recs):
            dec = decs[i % 2]
            dec.begin(d, 2 * n, enc, 0, n)          # reset: a new recording
            dec.exchange()
            if prev is not None:
                counts[prev[0]] = prev[1].finish()
                scan_ms.append(prev[1].ctx.last_scan_ms())
            prev = (i, dec)
        counts[prev[0]] = prev[1].finish()
        scan_ms.append(prev[1].ctx.last_scan_ms())

    def fence():
        for d in decs:                 # frame lists still on their way into the
            d.wait_frames()            # node's shared segment (sharded.py)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    fence()
    scan_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = float(np.mean(scan_ms))
    ms_per_step = dt / args.steps * 1e3
    total = world * R * n
    line = {
        "metric": "Msamples/s ADS-B decoded (2MS/s uc8+sc8 IQ, archive)",
        "value": round(total * args.steps / dt / 1e6, 1), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": (f"side measurement, configs[4]: archive reprocessing, {R} independent "
                         f"recordings of {n} samples per GPU (uc8 and sc8 alternating, a seed "
                         "each), sora_adsb_reset() + scan + resolve + frame list to host per "
                         "recording; a step is one pass over the recordings of a GPU"),
            "parallelism": "replicas only: every rank its own recordings, no collective",
            "recordings_per_gpu": R, "samples_per_recording": n,
            "frames_out_rank0": [int(c) for c in counts],
            "injected_frames_rank0": [int(r[2]) for r in recs],
            "list_regrows_rank0": [int(c.list_regrows()) for c in ctxs],
            "ms_per_recording": round(ms_per_step / R, 4),
        },
        "roofline": {
            "bound": "hbm", "achieved": round(2 * n / (kernel_ms * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(2 * n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "step_frac": round(2 * n * R / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernel": "adsb_scan_direct_kernel<256> (uc8, sc8)", "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes": 2 * n,
        },
        "cpu_baseline": None,
    }
    if rank == 0 and not args.no_cpu and args.cpu_samples > 0:
        # recording 0 (uc8) whole through the compiled reference: the same frames?
        d, enc, _ = recs[0]
        line["cpu_baseline"] = cpu_baseline(d, n, counts[0])
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for c in ctxs:
        c.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    args = parse_args()
    # stdout carries the one JSON line and nothing else: libraries that print
    # to fd 1 (RCCL's version banner does) go to stderr for the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.archive > 0:
        return run_archive(args, json_fd)
    import torch
    import torch.distributed as dist

    import sora_amd
    from sora_amd import FRAME_DTYPE, AdsbContext, synth
    from sora_amd._lib import BYTES_PER_SAMPLE as BPS, ENCODING_NAMES
    from sora_amd.sharded import LocalDecoder, ShardedDecoder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_collective:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=dev)              # RCCL over xGMI

    enc = ENCODING_NAMES[args.encoding]
    bps = BPS[enc]
    if args.scaling == "strong":
        from sora_amd.sharded import plan_shards
        total = int(args.samples)
        first, per_gpu, n_local = plan_shards(total, world)[rank]   # starts: multiples of 8
    else:
        per_gpu = int(args.samples)
        per_gpu -= per_gpu % 8                               # shard starts stay 16-byte aligned
        total = per_gpu * world
        first = rank * per_gpu
        n_local = per_gpu + (0 if rank == world - 1 else sora_amd.HALO)

    # ---- synthetic capture, generated straight into HBM ----
    params = synth.default_params(4 if args.encoding == "sc16" else 2, n_samples=total)
    params.encoding = enc
    frames = synth.make_frames(params)
    lo = np.searchsorted(frames["position"], max(0, first - 240))
    hi = np.searchsorted(frames["position"], first + n_local)
    d_iq = torch.empty(bps * n_local + 16, dtype=torch.uint8, device=dev)
    synth.fill_device(d_iq.data_ptr(), n_local, enc, params.seed, frames[lo:hi],
                      first=first)
    torch.cuda.synchronize()

    max_frames = max(per_gpu, int(args.samples) // world) // (
        64 if args.encoding == "sc16" else 1024) + 65536
    collective = world > 1 or args.force_collective
    # `inflight` decoders, each a context on a stream of its own: while one
    # capture's frames are resolved and copied to the host, the next capture's
    # scan already runs (two captures in flight is double buffering, the way an
    # archive of recordings is worked through).  Every step still does all of
    # its work inside the timed region.
    streams, rstreams, ctxs, decs = [], [], [], []
    two_stream = args.schedule == "two-stream" and args.inflight >= 2
    if two_stream:
        # the scans of all captures on one stream, in the order they are begun;
        # what follows a scan on the other (by default at high priority: its short
        # kernels take the CU slots the scan's workgroups free as they end)
        scan_stream = torch.cuda.Stream(device=dev, priority=0)
    for _ in range(max(1, args.inflight)):
        st = scan_stream if two_stream else torch.cuda.Stream(device=dev)
        # (a resolve stream per capture in flight: the host waits on it for the
        # frames of capture i-1 while capture i's resolve is already queued)
        resolve_stream = (torch.cuda.Stream(device=dev, priority=args.resolve_priority)
                          if two_stream else None)
        rstreams.append(resolve_stream)
        ctx = AdsbContext(n_local, device=local, max_squitters=max_frames,
                          max_frames=max_frames, stream=st.cuda_stream,
                          wire_frames=args.frames == "wire",
                          resolve_stream=resolve_stream.cuda_stream if two_stream else None)
        streams.append(st)                           # ~3x the injected frame rate
        ctxs.append(ctx)
        if collective:
            # scan -> squitter-record all-gather (RCCL) -> resolve -> every rank
            # copies its frames to its offset in a host segment shared by the node
            decs.append(ShardedDecoder(ctx, rank, world, dev, max_frames,
                                       stream=resolve_stream if two_stream else st,
                                       gather=args.gather))
        else:
            decs.append(LocalDecoder(ctx, max_frames))
    ctx = ctxs[0]
    job = (d_iq, bps * n_local, enc, first, per_gpu)

    scan_ms = []                       # HIP-event duration of every scan kernel launch

    phase_s = {"begin": 0.0, "finish": 0.0, "exchange": 0.0}

    def timed(name, fn, *a):
        t = time.perf_counter()
        r = fn(*a)
        phase_s[name] += time.perf_counter() - t
        return r

    def finish(d):
        n = timed("finish", d.finish)      # frames of that capture are on the host
        scan_ms.append(d.ctx.last_scan_ms())
        return n

    def mark(i):
        ev = torch.cuda.Event()
        ev.record(streams[i % len(decs)])
        return ev

    def run(k):
        """k steps, pipelined over the decoders; returns the last frame count.

        Kernels of different captures never share the GPU (measured again in
        round 2 with a scan at 3, 4 and 5 waves per SIMD: a scan that shares the
        CUs with a resolve takes as much longer as the resolve takes): events
        put them in the order S(i), R(i), S(i+1), R(i+1), ...  What overlaps a
        scan is the previous capture's frame copy to the host and the host's
        wait for it.  (--inflight 3 is the older schedule that moves the N>1
        exchange under the next scan.)"""
        n_dec, n = len(decs), 0
        if two_stream:
            # scans back to back on the one scan stream; the host stays n_dec - 1
            # captures ahead of the frames it waits for (with two decoders it has
            # what is left of a scan behind the previous resolve to queue the next
            # scan in; a short shard leaves it too little, a third decoder a step more)
            pending = []
            for i in range(k):
                d = decs[i % n_dec]
                timed("begin", d.begin, *job)
                timed("exchange", d.exchange)
                pending.append(d)
                if len(pending) == n_dec:
                    n = finish(pending.pop(0))
            while pending:
                n = finish(pending.pop(0))
            return n
        if n_dec >= 3:
            scanned, resolved = {}, {}
            for i in range(k + 2):
                if i < k:                                  # S(i) behind R(i-2)
                    if i - 2 in resolved:
                        streams[i % n_dec].wait_event(resolved.pop(i - 2))
                    timed("begin", decs[i % n_dec].begin, *job)
                    scanned[i] = mark(i)
                j = i - 1
                if 0 <= j < k:                             # R(j) behind S(j+1)
                    if j + 1 in scanned:
                        streams[j % n_dec].wait_event(scanned.pop(j + 1))
                    timed("exchange", decs[j % n_dec].exchange)
                    resolved[j] = mark(j)
                if 0 <= i - 2 < k:                         # frames of capture i-2
                    n = finish(decs[(i - 2) % n_dec])
            return n
        prev, resolved = None, None
        for i in range(k):
            d = decs[i % n_dec]
            if n_dec == 1 and prev is not None:
                n, prev = finish(prev), None
            if resolved is not None and not args.overlap and not two_stream:
                streams[i % n_dec].wait_event(resolved)    # S(i) behind R(i-1)
            timed("begin", d.begin, *job)  # reset (a new capture) + scan, asynchronous
            # (N>1: squitter exchange) + resolve, queued right behind the scan: the
            # GPU goes from one to the other without waiting for the host, which
            # meanwhile waits for the PREVIOUS capture's frames (until round 2 the
            # host queued this after that wait: 0.17 ms of idle GPU per step and
            # launch gaps between the small kernels, seen in the kernel trace)
            timed("exchange", d.exchange)
            if n_dec > 1 and not two_stream:
                resolved = mark(i)
            if prev is not None:
                n = finish(prev)
            prev = d
        if prev is not None:
            n = finish(prev)
        return n

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_frames = run(args.warmup)
    fence()
    scan_ms.clear()
    for k in phase_s:
        phase_s[k] = 0.0
    t0 = time.perf_counter()
    n_frames = run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    counts = ctx.counts()
    kernel_name = ctx.scan_kernel_name(enc)      # the kernel this context launched
    kernel_ms = float(np.mean(scan_ms))
    # With two streams a scan shares the GPU with the previous capture's resolve,
    # and its launch lasts longer for it: the roofline figure below is that
    # duration, as the timed region had it.  Beside it, behind the timed region,
    # the same launch with the GPU to itself (scans back to back, nothing else queued).
    alone = []
    for _ in range(3):
        for _ in range(8):      # back to back (the clock sags behind an idle GPU);
            ctx.reset()         # the events hold the last launch
            ctx.scan(*job[:3], base_position=first, n_owned=per_gpu)
        ctx.synchronize()
        alone.append(ctx.last_scan_ms())
    kernel_ms_alone = float(np.mean(alone))
    # HBM bytes per launch from the PMC passes of this same workload
    # (tools/hbm_traffic.py: FETCH_SIZE x2 + WRITE_SIZE); null if none matches
    traffic = traffic_source = None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            entry = json.load(fh).get(f"{kernel_name}:{args.encoding}:{n_local * bps}")
        if entry:
            traffic = round(entry["traffic_bytes"])
            traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / "
                              "WRITE_SIZE passes over this workload (tools/hbm_traffic.py), "
                              "committed; not re-measured in this run")
    except (OSError, ValueError):
        pass
    achieved = n_local * bps / (kernel_ms * 1e-3) / 1e9
    ms_per_step = dt / args.steps * 1e3
    # the whole step against the same roofline: algorithmic bytes of all ranks
    # over the step time, per GPU
    step_gbs = total * bps / world / (ms_per_step * 1e-3) / 1e9

    hour = args.encoding == "uc8" and int(args.samples) == 7200000000
    if hour and world == 1:
        workload = ("configs[1]: 1-hour synthetic 2 MS/s uc8 capture (DF17/DF11 at mixed "
                    "SNR), scan + resolve + frame list to host")
    elif hour and args.scaling == "strong":
        workload = (f"configs[2]: the same 1-hour synthetic 2 MS/s uc8 capture in {world} "
                    "haloed time shards, scan + squitter exchange + resolve + frame lists "
                    "to host")
    elif hour:
        workload = (f"side measurement (weak scaling): a 1-hour uc8 shard per GPU of a "
                    f"{world}-hour capture")
    else:
        workload = (f"side measurement: {total} samples of {args.encoding} over {world} "
                    "GPU(s), scan + resolve + frame list to host")
    # what "frame list to host" is in this run (the advisor's note on round 2: the
    # 28-byte form is a different deliverable from the 48-byte struct and the line
    # must say which one it timed)
    workload += (" (28-byte wire frames, SORA_OPT_WIRE_FRAMES: bits, position, amplitude, "
                 "flags; DF / syndrome / address are derived from the bits on demand by "
                 "sora_adsb_unpack_frames, outside the step)" if args.frames == "wire" else
                 " (48-byte struct sora_adsb_frame)")
    line = {
        "metric": f"Msamples/s ADS-B decoded (2MS/s {args.encoding} IQ)",
        "value": round(total * args.steps / dt / 1e6, 1),
        "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32" if args.encoding == "sc16" else "u8", "data": "synthetic",
        "config": {
            "workload": workload,
            "parallelism": "1 GPU" if world == 1 else
                           f"{world} haloed time shards, one rank per GPU ({args.scaling})",
            "exchange": None if not collective else
                        "squitter records: one 64 KB all-gather over RCCL/xGMI per step",
            "gather": None if not collective else
                      ("frame lists to a host segment shared by the node's ranks, every rank "
                       "over its own PCIe link (--gather shm)" if decs[0].gather == "shm" else
                       "frame lists gathered to rank 0 with RCCL over xGMI (--gather collective)"),
            "samples_total": total,
            "samples_per_gpu": per_gpu, "bytes_per_gpu": bps * n_local,
            "injected_frames": int(len(frames)), "frames_out": int(n_frames),
            "candidates_rank0": int(counts["candidates"]),
            "captures_in_flight": len(decs),
            "schedule": ("two streams: scans back to back on one, exchange + resolve + frame "
                         f"copy on the other (priority {args.resolve_priority})" if two_stream
                         else "a stream per capture in flight, S(i) R(i) S(i+1) ordered by events"),
            "frame_bytes": int(ctx.frame_dtype.itemsize),
            "host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in phase_s.items()},
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": traffic_source,
            "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
            "step_frac_is": "algorithmic bytes per GPU / ms_per_step / peak: the whole "
                            "step (scan + resolve + frames to host), not the scan kernel alone",
            "kernel": f"{kernel_name} ({args.encoding})",
            "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": n_local * bps,
            "kernel_ms_unshared": round(kernel_ms_alone, 4),
            "frac_unshared": round(n_local * bps / (kernel_ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "unshared_is": "the same launch with the GPU to itself (behind the timed region); "
                           "`achieved` / `frac` are the launches of the timed region, which "
                           + ("share the GPU with the previous capture's resolve" if two_stream
                              else "run alone as well (events schedule)"),
        },
    }

    if (rank == 0 and world == 1 and not args.no_cpu and args.cpu_samples > 0
            and args.encoding == "uc8"):          # the reference reads uc8 only
        n_cpu = min(int(args.cpu_samples), per_gpu)
        last = decs[(args.steps - 1) % len(decs)].frames()
        pos = (sora_amd.unpack_frames(last, enc) if args.frames == "wire" else last)["position"]
        in_window = int(np.count_nonzero(pos < n_cpu - 239))
        line["cpu_baseline"] = cpu_baseline(d_iq, n_cpu, in_window)
    else:
        line["cpu_baseline"] = None

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for d in decs:
        d.close()
    for c in ctxs:
        c.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
