"""The N>1 path (sora_amd/sharded.py) with world_size 2 over gloo.

CPU tier: the per-rank device context is replaced by an oracle-backed stand-in
with the same interface (tests may use the oracle; the product never does), so
what is tested is what sharded.py owns: the shard plan and its halos, the
squitter record exchange, the first-seen rule across shards, and the
rank-ordered gather.

GPU tier (-m gpu): the same two processes, each with a REAL AdsbContext on the
one GPU of the box and the real ShardedDecoder(gather="shm") -- the /dev/shm
header and spin protocol, the pinned staging copy, two decoders interleaved the
way bench.py pipelines them -- against the whole-capture oracle list.  RCCL
refuses two ranks on one device, so the 64 KB exchange block travels over gloo
(staged through pinned host memory, sharded.py); everything else is the code
that runs with one rank per GPU.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from sora_amd import FRAME_ACCEPTED, FRAME_DTYPE, FRAME_SQUITTER, string_to_bits, synth
from sora_amd.sharded import ShardedDecoder, plan_shards


class OracleContext:
    """AdsbContext's interface on CPU tensors, computed by the oracle."""

    def __init__(self):
        self.known = {}                 # address -> first global position

    def reset(self):
        self.known = {}

    def scan(self, d_iq, nbytes, encoding, base_position=0, n_owned=None):
        raw = d_iq[:nbytes].numpy().tobytes()
        # every candidate, judged later against first-seen positions
        self.cands = [c for c in ob.scan(raw, encoding, base_position=base_position)
                      if c["position"] < base_position + n_owned]
        self.squitters = [c for c in self.cands
                          if c["df"] in (11, 17, 18) and c["syndrome"] == 0]

    def squitter_count(self):
        return len(self.squitters)

    def copy_squitters(self, d_dst, cap):
        rec = [(c["position"] << 24) | c["address"] for c in self.squitters][:cap]
        d_dst[:len(rec)] = torch.tensor(rec, dtype=torch.int64)

    def export_block(self, d_block, block_records):
        first = {}
        for c in self.squitters:
            first[c["address"]] = min(first.get(c["address"], 1 << 62), c["position"])
        rec = [(pos << 24) | a for a, pos in sorted(first.items())]
        d_block[0] = len(rec)
        rec = rec[:block_records - 1]
        d_block[1:1 + len(rec)] = torch.tensor(rec, dtype=torch.int64)

    def add_known_blocks(self, d_blocks, n_blocks, block_records, skip_block=-1):
        for b in range(n_blocks):
            if b == skip_block:
                continue
            blk = d_blocks[b * block_records:(b + 1) * block_records]
            n = min(int(blk[0]), block_records - 1)
            self.add_known(blk[1:], n)

    def synchronize(self):
        pass

    def _note(self, address, position):
        if position < self.known.get(address, 1 << 62):
            self.known[address] = position

    def add_known(self, d_records, n):
        for r in d_records[:n].tolist():
            self._note(r & 0xFFFFFF, r >> 24)

    def resolve(self):
        for c in self.squitters:
            self._note(c["address"], c["position"])
        out = []
        for c in self.cands:
            squitter = c["df"] in (11, 17, 18)
            ok = (c["syndrome"] == 0) if squitter else \
                self.known.get(c["address"], 1 << 62) < c["position"]
            if ok:
                out.append((c, squitter))
        self.frames = np.zeros(len(out), dtype=FRAME_DTYPE)
        for i, (c, squitter) in enumerate(out):
            self.frames[i] = (c["position"], tuple(string_to_bits(c["bits"])),
                              c["amplitude_56"], c["syndrome"], c["address"], 0, c["df"],
                              -1, FRAME_ACCEPTED | (FRAME_SQUITTER if squitter else 0), 0)

    def frame_count(self):
        return len(self.frames)

    def collect_into(self, out):
        out[:len(self.frames)] = self.frames
        return len(self.frames)

    def collect_device(self, d_out, cap):
        raw = torch.from_numpy(self.frames.view(np.uint8).copy())
        d_out[:raw.numel()] = raw
        return len(self.frames)


def _worker(rank, world, port, raw, result_path, gather, exchange_records=64, gpu=False,
            max_frames=20000, ctx_kw=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = len(raw) // 2
        first, owned, local = plan_shards(total, world)[rank]
        shard = torch.frombuffer(bytearray(raw[2 * first:2 * (first + local)] + bytes(16)),
                                 dtype=torch.uint8)
        # two decoders interleaved the way bench.py pipelines them
        if gpu:
            from sora_amd import AdsbContext
            torch.cuda.set_device(0)
            dev = torch.device("cuda", 0)
            shard = shard.to(dev)
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            ctxs = [AdsbContext(local, device=0, stream=st.cuda_stream, **(ctx_kw or {}))
                    for st in streams]
            decs = [ShardedDecoder(c, rank, world, dev, max_frames, gather=gather, stream=st,
                                   exchange_records=exchange_records)
                    for c, st in zip(ctxs, streams)]
        else:
            decs = [ShardedDecoder(OracleContext(), rank, world, torch.device("cpu"),
                                   max_frames, gather=gather,
                                   exchange_records=exchange_records)
                    for _ in range(2)]
        args = (shard, 2 * local, ob.ENC_UC8, first, owned)
        decs[0].begin(*args)
        decs[0].exchange()
        decs[1].begin(*args)
        n0 = decs[0].finish()
        decs[1].exchange()
        decs[0].begin(*args)                    # again: reset between captures works
        n1 = decs[1].finish()
        decs[0].exchange()
        n = decs[0].finish()
        assert n0 == n1 == n
        if rank == 0:
            assert np.array_equal(decs[0].frames(), decs[1].frames())
            assert len(decs[0].frames()) == n
            np.save(result_path, decs[0].frames().copy())
        if gpu:
            with open(f"{result_path}.{rank}.regrows", "w") as fh:
                fh.write(str(sum(c.list_regrows() for c in ctxs)))
        for d in decs:
            d.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _same_as_oracle(got, want):
    from conftest import assert_same_frames, frames_as_dicts
    assert_same_frames(frames_as_dicts(got), want)


def test_plan_shards():
    plan = plan_shards(1000003, 4)
    assert sum(p[1] for p in plan) == 1000003
    assert all(p[0] % 8 == 0 for p in plan)
    assert [p[2] - p[1] for p in plan] == [239, 239, 239, 0]
    assert plan_shards(100, 8)[-1] == (100, 0, 0)       # more ranks than work


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gather", ["shm", "collective"])
def test_two_ranks_equal_one(tmp_path, gather):
    p = synth.default_params(2, n_samples=300000, mean_gap=900, reply_permille=300,
                             unknown_permille=250, n_icao=5, seed=21)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    # the stateful rule must cross the shard boundary in this capture
    half = plan_shards(p.n_samples, 2)[1][0]
    late = [c for c in want if c["position"] >= half and c["df"] not in (11, 17, 18)]
    assert late, "test capture has no reply in the second shard"

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(2, port, raw, out, gather), nprocs=2, join=True)
    got = np.load(out)
    assert [int(f["position"]) for f in got] == [c["position"] for c in want]
    from sora_amd import bits_to_string
    assert [bits_to_string(f["bits"]) for f in got] == [c["bits"] for c in want]
    assert [float(f["amplitude_56"]) for f in got] == [c["amplitude_56"] for c in want]


def test_local_decoder_phases():
    """LocalDecoder is ShardedDecoder's three phases with no exchange: begin scans,
    exchange resolves, finish collects (bench.py runs N=1 through it)."""
    from sora_amd.sharded import LocalDecoder
    p = synth.default_params(2, n_samples=60000, mean_gap=900, n_icao=3, seed=3)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    ctx = OracleContext()
    ctx.collect_into = lambda out: (out.__setitem__(slice(0, len(ctx.frames)), ctx.frames),
                                    len(ctx.frames))[1]
    dec = LocalDecoder(ctx, 5000, pin=False)
    shard = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    for _ in range(2):
        n = dec.step(shard, len(raw), ob.ENC_UC8, 0, p.n_samples)
    want = [c for c in ob.scan(raw) if c["accepted"]]
    assert n == len(want) > 10
    assert [int(f["position"]) for f in dec.frames()] == [c["position"] for c in want]


def _overflow_worker(rank, world, port, raw, path, gpu=False):
    try:
        _worker(rank, world, port, raw, path, "shm", exchange_records=3, gpu=gpu)
    except RuntimeError as e:
        assert "exchange records exceed" in str(e)
        with open(f"{path}.{rank}.raised", "w") as fh:
            fh.write(str(e))


@pytest.mark.timeout(300)
def test_exchange_block_overflow_is_reported(tmp_path):
    """A block too small for the addresses a shard saw first must not pass
    silently: every rank raises when it collects its frames."""
    p = synth.default_params(2, n_samples=200000, mean_gap=900, n_icao=5, seed=21)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frames.npy")
    mp.spawn(_overflow_worker, args=(2, port, raw, out), nprocs=2, join=True)
    assert os.path.exists(out + ".0.raised") and os.path.exists(out + ".1.raised")


def _failing_worker(rank, world, port, raw, path):
    """Rank 1's context fails when its frames are asked for; rank 0 must not spin
    until the timeout."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = len(raw) // 2
        first, owned, local = plan_shards(total, world)[rank]
        shard = torch.frombuffer(bytearray(raw[2 * first:2 * (first + local)] + bytes(16)),
                                 dtype=torch.uint8)
        ctx = OracleContext()
        if rank == 1:
            def boom(*_a):
                raise RuntimeError("7 frames exceed max_frames")
            ctx.collect_device = boom
        dec = ShardedDecoder(ctx, rank, world, torch.device("cpu"), 20000, gather="shm")
        dec.begin(shard, 2 * local, ob.ENC_UC8, first, owned)
        dec.exchange()
        t0 = time.monotonic()
        try:
            dec.finish()
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        with open(f"{path}.{rank}", "w") as fh:
            fh.write(f"{time.monotonic() - t0:.3f}\n{msg}")
        dec.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_a_failing_rank_fails_the_others_at_once(tmp_path):
    """An error on one rank between begin() and finish() is marked in the shared
    header; the ranks waiting for that rank's frames raise within a second
    instead of spinning for SPIN_TIMEOUT_S."""
    p = synth.default_params(2, n_samples=120000, mean_gap=900, n_icao=4, seed=8)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    out = str(tmp_path / "fail")
    mp.spawn(_failing_worker, args=(2, _free_port(), raw, out), nprocs=2, join=True)
    t1, msg1 = open(out + ".1").read().split("\n", 1)
    assert "exceed max_frames" in msg1
    t0, msg0 = open(out + ".0").read().split("\n", 1)
    assert "rank 1 failed" in msg0 and float(t0) < 5.0


def four_shard_capture():
    """800 k samples for four ranks with the stateful rule crossing shard edges
    both ways: address A squitters first in shard 0 and is replied to in shards 0
    (before: rejected), 1 and 3 (accepted); address B squitters first in shard 2 and
    has replies in shards 0, 1 and early in shard 2 (all rejected: plane.c:479-483
    knows B only from its squitter on) and later in shards 2 and 3 (accepted)."""
    A, B = 0x4840D6, 0xABC123
    sq = lambda icao, me: synth.build_df17(icao, me)            # noqa: E731
    rp = lambda icao, k: synth.build_short_reply(4, 0x123400 + k, icao)     # noqa: E731
    rows = [
        (10000, rp(A, 1), 70, 0, 56),       # before A's first squitter: rejected
        (50000, sq(A, 0x58B986D0B3BD25), 70, 0, 112),
        (60000, rp(A, 2), 70, 0, 56),       # shard 0, accepted
        (120000, rp(B, 3), 70, 0, 56),      # B unknown yet: rejected
        (250000, rp(A, 4), 70, 0, 56),      # shard 1: A known from shard 0
        (300000, rp(B, 5), 70, 0, 56),      # shard 1: B is first seen in shard 2: rejected
        (420000, rp(B, 6), 70, 0, 56),      # shard 2, before B's squitter: rejected
        (450000, sq(B, 0x99440994083817), 70, 0, 112),
        (480000, rp(B, 7), 70, 0, 56),      # shard 2, accepted
        (520000, sq(A, 0x58B986D0B3BD25), 70, 0, 112),   # a later squitter of A changes nothing
        (650000, rp(A, 8), 70, 0, 56),      # shard 3: A from shard 0
        (700000, rp(B, 9), 70, 0, 56),      # shard 3: B from shard 2
        (799000, synth.build_long_reply(20, 0x0a1b2c, 0x11223344556677, A), 70, 0, 112),
    ]
    raw = synth.fill_host(800000, ob.ENC_UC8, 4242, synth.frame_table(rows)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    got = {c["position"] for c in want}
    assert {50000, 60000, 250000, 450000, 480000, 520000, 650000, 700000, 799000} <= got
    assert not ({10000, 120000, 300000, 420000} & got)
    return raw, want


@pytest.mark.timeout(300)
def test_four_ranks_first_seen_crosses_two_boundaries(tmp_path):
    raw, want = four_shard_capture()
    assert [p[0] for p in plan_shards(len(raw) // 2, 4)] == [0, 200000, 400000, 600000]
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(4, _free_port(), raw, out, "shm"), nprocs=4, join=True)
    _same_as_oracle(np.load(out), want)


# ---- GPU tier: two processes, real contexts, one device ----------------------


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("gather", ["shm", "collective"])
def test_two_gpu_processes_equal_the_whole_capture(tmp_path, gather):
    """BASELINE.json configs[2] with two ranks: haloed time shards, the squitter
    exchange, every rank's frames into the node's host segment, rank 0 reads the
    position-sorted whole; bit-exact with the oracle on the unsharded capture."""
    p = synth.default_params(2, n_samples=1600000, mean_gap=700, reply_permille=300,
                             unknown_permille=250, n_icao=9, seed=23)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    half = plan_shards(p.n_samples, 2)[1][0]
    # addresses first seen in shard 0 whose replies lie in shard 1: the rule
    # crosses the boundary
    assert [c for c in want if c["position"] >= half and c["df"] not in (11, 17, 18)]
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(2, _free_port(), raw, out, gather, 64, True), nprocs=2, join=True)
    _same_as_oracle(np.load(out), want)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_gpu_processes_small_lists_and_block_overflow(tmp_path):
    """Record and frame lists far too small on both ranks: each rank re-runs its
    shard locally (the squitter export was whole), the result is still exact.
    And an exchange block too small for the addresses raises on BOTH ranks."""
    p = synth.default_params(2, n_samples=900000, mean_gap=700, reply_permille=300,
                             unknown_permille=250, n_icao=6, seed=29)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(2, _free_port(), raw, out, "shm", 64, True, 20000,
                            {"max_candidates": 16, "max_frames": 16, "wire_frames": True}),
             nprocs=2, join=True)
    import sora_amd                      # 28-byte wire frames through the segment
    got = np.load(out)
    assert got.dtype.itemsize == 28
    _same_as_oracle(sora_amd.unpack_frames(got, ob.ENC_UC8), want)
    assert all(int(open(f"{out}.{r}.regrows").read()) >= 2 for r in range(2))
    out2 = str(tmp_path / "frames2.npy")
    mp.spawn(_overflow_worker, args=(2, _free_port(), raw, out2, True), nprocs=2, join=True)
    assert os.path.exists(out2 + ".0.raised") and os.path.exists(out2 + ".1.raised")


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("gather", ["shm", "collective"])
def test_four_gpu_processes_first_seen_crosses_two_boundaries(tmp_path, gather):
    """World 4 on the one GPU (four processes, eight contexts): an address first
    seen in shard 0 with replies in shard 3, and one first seen in shard 2 whose
    earlier replies in shards 0-2 stay rejected; bit-exact with the oracle."""
    raw, want = four_shard_capture()
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(4, _free_port(), raw, out, gather, 64, True), nprocs=4, join=True)
    _same_as_oracle(np.load(out), want)
