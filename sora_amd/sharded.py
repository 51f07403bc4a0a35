"""Time-sharded decoding over several GPUs.

One process per GPU; rank r owns the positions [r*per, (r+1)*per) of the
capture and holds 239 halo samples behind them.  The scan needs no
communication.  The path's one exchange: per address a rank saw first, its
earliest clean squitter (position << 24 | address, 8 bytes; a few hundred
records per hour) is all-gathered so that each rank's first-seen table knows
the addresses earlier ranks introduced
(`torch.distributed`: backend nccl = RCCL over xGMI on the GPUs, gloo in the
CPU tests).  It is ONE all-gather of a fixed-size block per rank (count, then
records), queued on the context's stream between the scan and the resolve:
the counts are read on the device, so the host never waits inside a step.
Sample data never moves.

The frame lists go to the host, not to another GPU: the ranks of one node
share a host-memory segment (/dev/shm) with one slot per rank, and every rank
copies its frames over ITS OWN PCIe link into its slot (by DMA if the segment
can be page-locked, else by DMA into a pinned buffer of the rank and a CPU copy
from there, which a helper thread does while the rank goes on to its next
capture: finish() returns once the COUNTS of all ranks are known, frames() and
wait_frames() wait for the lists).  Counts and completion travel through a small
header of the same segment, so this phase issues no device collective: one
would queue behind the other capture's scan kernel, whose persistent
workgroups leave no room on the CUs until it ends.  The slots in rank order
are the position order; rank 0 reads the whole list there without a byte of
it crossing xGMI or queueing on one PCIe link.  `gather="collective"` keeps
the plain RCCL gather to rank 0 (ranks on several nodes).

A step is three phases so that two decoders (two contexts, two streams) can be
interleaved: begin() launches the scan, exchange() queues the export, the
all-gather and the resolve behind it, finish() waits for the frames and places
them.  step() runs the three in a row.
"""
import concurrent.futures
import contextlib
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import BYTES_PER_SAMPLE, FRAME_DTYPE, HALO


def _frame_dtype(ctx):
    """What the context hands out: 48-byte frames, or 28-byte wire frames under
    SORA_OPT_WIRE_FRAMES (sora_amd.unpack_frames turns them into the former)."""
    return getattr(ctx, "frame_dtype", FRAME_DTYPE)

# u64 per rank: seq of the frames in the slot, n_frames, ack (rank 0), failed,
# seq n_frames belongs to, pad
HEADER_WORDS = 8
EXCHANGE_RECORDS = 8192             # words of a rank's exchange block: count + records
SPIN_TIMEOUT_S = 300.0


def plan_shards(total_samples, world):
    """[(first, n_owned, n_local)] per rank; shard starts are multiples of 8
    samples so that every shard of one device buffer stays 16-byte aligned."""
    per = -(-total_samples // world)
    per += -per % 8
    out = []
    for r in range(world):
        first = min(r * per, total_samples)
        owned = min(per, total_samples - first)
        local = min(owned + HALO, total_samples - first)
        out.append((first, owned, local))
    return out


class LocalDecoder:
    """The same three phases on one GPU with no exchange at all."""

    def __init__(self, ctx, max_frames, pin=True):
        self.ctx = ctx
        self.dtype = _frame_dtype(ctx)
        self.h_pin = torch.empty(int(max_frames) * self.dtype.itemsize, dtype=torch.uint8,
                                 pin_memory=pin)
        self.h_frames = self.h_pin.numpy().view(self.dtype)
        self.n = 0

    def begin(self, d_iq, nbytes, encoding, base_position, n_owned, reset=True):
        if reset:
            self.ctx.reset()
        self.ctx.scan(d_iq, nbytes, encoding, base_position=base_position,
                      n_owned=n_owned)

    def exchange(self):
        self.ctx.resolve()

    def finish(self):
        self.n = self.ctx.collect_into(self.h_frames)
        return self.n

    def wait_frames(self):
        pass

    def frames(self):
        return self.h_frames[:self.n]

    def step(self, *args, **kw):
        self.begin(*args, **kw)
        self.exchange()
        return self.finish()

    def close(self):
        pass


class ShardedDecoder:
    """Drives one context per rank through scan -> exchange -> resolve -> gather.

    `stream`: the torch.cuda.Stream the context queues its exchange and resolve
    on -- the one it was created on (AdsbContext(..., stream=s.cuda_stream)), or
    its resolve stream if it has one (resolve_stream=s.cuda_stream); collectives
    and context launches are then ordered by that one stream.  None: the context
    must be on torch's current stream, or be a CPU stand-in."""

    def __init__(self, ctx, rank, world, device, max_frames, group=None, stream=None,
                 gather="shm", exchange_records=EXCHANGE_RECORDS):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.dtype = _frame_dtype(ctx)
        self.frame_bytes = self.dtype.itemsize
        self.device, self.max_frames, self.group = device, int(max_frames), group
        self.stream, self.gather = stream, gather
        self.cuda = device.type == "cuda"
        # exchange block: word 0 the count, then one record per address this shard
        # saw first (not one per squitter)
        self.block = max(2, int(exchange_records))
        self.d_blk = torch.zeros(self.block, dtype=torch.int64, device=device)
        self.d_blk_all = torch.zeros(world * self.block, dtype=torch.int64, device=device)
        # gloo moves host memory: with device contexts under a gloo group (several
        # ranks sharing one GPU, which RCCL refuses; tests) the 64 KB block is
        # staged through pinned host memory at the cost of one stream wait
        self.stage_exchange = self.cuda and dist.get_backend(group) == "gloo"
        if self.stage_exchange:
            self.h_blk = torch.zeros(self.block, dtype=torch.int64, pin_memory=True)
            self.h_blk_all = torch.zeros(world * self.block, dtype=torch.int64,
                                         pin_memory=True)
        self.h_exported = torch.zeros(world, dtype=torch.int64, pin_memory=self.cuda)
        self.counts = torch.zeros(world, dtype=torch.int64, device=device)
        self.d_out = torch.zeros(self.max_frames * self.frame_bytes, dtype=torch.uint8,
                                 device=device)
        self.seq = 0
        self.ns = [0] * world
        self._registered = None
        self._mm = None
        self.header = None
        self._copier = self._copy = None
        if gather == "shm" and not self._open_segment():
            self.gather = gather = "collective"  # no room in /dev/shm on this node
        if gather != "shm":
            self.d_gather = ([torch.zeros_like(self.d_out) for _ in range(world)]
                             if rank == 0 else None)
            n_host = self.max_frames * (world if rank == 0 else 1)
            self.h_pin = torch.empty(n_host * self.frame_bytes, dtype=torch.uint8,
                                     pin_memory=self.cuda)
            self.h_frames = self.h_pin.numpy().view(self.dtype)

    # -- the shared host segment -------------------------------------------
    def _open_segment(self):
        self.slot_bytes = max(1, self.max_frames) * self.frame_bytes
        self.slot_bytes += -self.slot_bytes % 4096
        self.data_at = 4096                      # the header has a page of its own
        nbytes = self.data_at + self.world * self.slot_bytes
        nbytes += -nbytes % (2 << 20)
        name = [None]
        if self.rank == 0:
            try:
                vfs = os.statvfs("/dev/shm")
                if vfs.f_bavail * vfs.f_frsize > 2 * nbytes:
                    name[0] = f"/dev/shm/sora_frames_{os.getpid()}_{id(self):x}"
                    with open(name[0], "wb") as fh:
                        fh.truncate(nbytes)      # zero filled: seq = ack = 0
            except OSError:
                name[0] = None
        dist.broadcast_object_list(name, src=0, group=self.group)
        if name[0] is None:
            return False
        self._mm = np.memmap(name[0], dtype=np.uint8, mode="r+", shape=(nbytes,))
        dist.barrier(group=self.group)           # every rank has it mapped
        if self.rank == 0:
            os.unlink(name[0])                   # the mappings keep it alive
        self.header = self._mm[:self.world * HEADER_WORDS * 8].view(np.uint64).reshape(
            self.world, HEADER_WORDS)
        self.slots = [self._mm[self.data_at + r * self.slot_bytes:
                               self.data_at + r * self.slot_bytes +
                               self.max_frames * self.frame_bytes].view(self.dtype)
                      for r in range(self.world)]
        if self.cuda:
            idx = self.device.index if self.device.index is not None else 0
            rc = _lib.lib().sora_adsb_host_register(idx, self._mm.ctypes.data, nbytes)
            if rc == _lib.OK:
                self._registered = (idx, self._mm.ctypes.data)
        if self._registered is None:
            # the segment cannot be page-locked here (RLIMIT_MEMLOCK, containers):
            # DMA into a pinned buffer of this rank, then a CPU copy into the slot
            self.h_stage_t = torch.empty(self.max_frames * self.frame_bytes, dtype=torch.uint8,
                                         pin_memory=self.cuda)
            self.h_stage = self.h_stage_t.numpy().view(self.dtype)
            if self.cuda:
                self._copier = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            base = self.data_at + self.rank * self.slot_bytes
            self.h_slot_t = torch.from_numpy(self._mm)[base:base + self.slot_bytes]
        return True

    def close(self):
        try:
            self._join_copy()
        finally:
            if self._copier is not None:
                self._copier.shutdown(wait=True)
                self._copier = None
            self._close_segment()

    def _close_segment(self):
        if self._registered is not None:
            _lib.lib().sora_adsb_host_unregister(*self._registered)
            self._registered = None
        self.header = self.slots = self.h_slot_t = None
        self._mm = None

    def _on_stream(self):
        if self.stream is not None:
            return torch.cuda.stream(self.stream)
        return contextlib.nullcontext()

    def _all_counts(self, k):
        dev = "cpu" if self.stage_exchange else self.device
        mine = torch.tensor([k], dtype=torch.int64, device=dev)
        counts = self.counts.cpu() if self.stage_exchange else self.counts
        dist.all_gather_into_tensor(counts, mine, group=self.group)
        return [int(x) for x in counts.tolist()]

    def _spin(self, ready, what):
        """Waits for another rank's word in the header -- or for its failure
        mark: a rank that raises anywhere between begin() and finish() says so
        in the header first (_failing), and the ranks that wait for it raise at
        once instead of spinning until the timeout."""
        t0 = time.monotonic()
        while not ready():
            failed = np.flatnonzero(self.header[:, 3])
            if failed.size:
                raise RuntimeError(f"rank {self.rank}: rank {int(failed[0])} failed while this "
                                   f"rank waited for {what}")
            if time.monotonic() - t0 > SPIN_TIMEOUT_S:
                raise RuntimeError(f"rank {self.rank}: timed out waiting for {what}")
            time.sleep(0)

    @contextlib.contextmanager
    def _failing(self):
        """Whatever is raised inside is marked in the shared header before it
        goes on up (the mark stays: the segment is not usable afterwards)."""
        try:
            yield
        except BaseException:
            if self.gather == "shm" and self.header is not None:
                self.header[self.rank, 3] = self.seq + 1
            raise

    # -- the three phases ---------------------------------------------------
    def begin(self, d_iq, nbytes, encoding, base_position, n_owned, reset=True):
        if self.gather == "shm" and self.rank == 0:
            # the reader lets go of the previous result of this decoder
            self.header[0, 2] = self.seq
        self.seq += 1
        with self._failing():
            if reset:
                self.ctx.reset()
            self.ctx.scan(d_iq, nbytes, encoding, base_position=base_position,
                          n_owned=n_owned)

    def exchange(self):
        ctx, world = self.ctx, self.world
        with self._failing(), self._on_stream():
            ctx.export_block(self.d_blk, self.block)
            if self.stage_exchange:
                self.h_blk.copy_(self.d_blk, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                dist.all_gather_into_tensor(self.h_blk_all, self.h_blk, group=self.group)
                self.d_blk_all.copy_(self.h_blk_all, non_blocking=True)
            else:
                dist.all_gather_into_tensor(self.d_blk_all, self.d_blk, group=self.group)
            ctx.add_known_blocks(self.d_blk_all, world, self.block, skip_block=self.rank)
            ctx.resolve()
            # what every rank exported, for the overflow check in finish()
            self.h_exported.copy_(self.d_blk_all.view(world, self.block)[:, 0],
                                  non_blocking=True)

    def _check_exported(self):
        """Behind the wait for the frames: a rank that saw more new addresses than
        its block holds dropped some, and every rank's result is then wrong."""
        most = int(self.h_exported.max())
        if most > self.block - 1:
            raise RuntimeError(f"{most} exchange records exceed the block of "
                               f"{self.block - 1}: raise exchange_records")

    def finish(self):
        """Total number of frames (all ranks); on rank 0, frames() returns them in
        position order (valid until this decoder's next begin())."""
        with self._failing(), self._on_stream():
            if self.gather == "shm":
                return self._finish_shm()
            return self._finish_collective()

    def _finish_shm(self):
        ctx, r = self.ctx, self.rank
        # rank 0 has let go of what this slot held (two steps ago)
        self._spin(lambda: int(self.header[0, 2]) >= self.seq - 1, "the reader's release")
        if self._registered is not None:
            n = ctx.frame_count()
            if n > self.max_frames:
                raise RuntimeError(f"{n} frames exceed max_frames")
            if n:
                assert ctx.collect_into(self.slots[r][:n]) == n      # DMA into the slot
        else:
            # DMA into this rank's pinned buffer; from there into the slot by the
            # CPU (non-temporal stores: the ranks of a node copy at the same time),
            # on a helper thread: 0.2 ms for 6 MB that the rank's main thread has
            # better use for -- it is the one that queues the next capture's scan
            self._join_copy()                    # the buffer is free again
            n = ctx.collect_into(self.h_stage) if self.cuda else self._stand_in_frames()
            if n > self.max_frames:
                raise RuntimeError(f"{n} frames exceed max_frames")
        self._check_exported()
        # publish: the count and the step it belongs to ...
        self.header[r, 1] = n
        self.header[r, 4] = self.seq
        # ... and, once the frames are in the slot, the sequence number their
        # readers wait for
        if self._registered is not None:
            self.header[r, 0] = self.seq
        elif self._copier is not None:
            self._copy = self._copier.submit(self._place, n, self.seq)
        else:
            self._place(n, self.seq)
        self._spin(lambda: bool((self.header[:, 4] >= self.seq).all()), "the other ranks")
        self.ns = [int(x) for x in self.header[:, 1]]
        return sum(self.ns)

    def _stand_in_frames(self):
        """CPU stand-ins for a context (tests) hand their frames out as a tensor."""
        n = self.ctx.collect_device(self.d_out, self.max_frames)
        nb = min(n, self.max_frames) * self.frame_bytes
        self.h_stage_t[:nb].copy_(self.d_out[:nb])
        return n

    def _place(self, n, seq):
        """Pinned buffer -> this rank's slot, then the word that says so."""
        try:
            nb = n * self.frame_bytes
            if self.cuda:
                _lib.check(_lib.lib().sora_adsb_host_copy(
                    self.h_slot_t.data_ptr(), self.h_stage_t.data_ptr(), nb))
            else:
                self.h_slot_t[:nb].copy_(self.h_stage_t[:nb])
            self.header[self.rank, 0] = seq
        except BaseException:
            self.header[self.rank, 3] = seq + 1  # as _failing(): the others stop waiting
            raise

    def _join_copy(self):
        copy, self._copy = self._copy, None
        if copy is not None:
            copy.result()                        # raises what the helper raised

    def wait_frames(self):
        """The frame lists of the last finish() are in place: this rank's in its
        slot and, on rank 0, every rank's."""
        with self._failing():
            self._join_copy()
            if self.gather == "shm" and self.rank == 0:
                self._spin(lambda: bool((self.header[:, 0] >= self.seq).all()),
                           "the other ranks' frames")

    def _finish_collective(self):
        ctx, world, rank = self.ctx, self.world, self.rank
        ns = self._all_counts(ctx.collect_device(self.d_out, self.max_frames))
        self._check_exported()
        if max(ns) > self.max_frames:
            raise RuntimeError(f"{max(ns)} frames exceed max_frames")
        self.ns = ns
        m = max(ns) * self.frame_bytes
        if m and self.stage_exchange:           # gloo: through host memory
            mine = self.d_out[:m].cpu()
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, parts, dst=0, group=self.group)
            for r in range(world if rank == 0 else 0):
                self.d_gather[r][:m].copy_(parts[r])
        elif m:
            dist.gather(self.d_out[:m],
                        [g[:m] for g in self.d_gather] if rank == 0 else None,
                        dst=0, group=self.group)
        if rank == 0:
            off = 0
            for r in range(world):
                nb = ns[r] * self.frame_bytes
                self.h_pin[off:off + nb].copy_(self.d_gather[r][:nb], non_blocking=True)
                off += nb
            if self.cuda:
                torch.cuda.current_stream().synchronize()
        return sum(ns)

    def frames(self):
        """The frame list of the last finish(), in position order (a copy when it
        has to be stitched from the per-rank slots)."""
        if self.gather != "shm":
            return self.h_frames[:sum(self.ns)]
        self.wait_frames()
        if self.world == 1:
            return self.slots[0][:self.ns[0]]
        return np.concatenate([self.slots[q][:self.ns[q]] for q in range(self.world)])

    def step(self, d_iq, nbytes, encoding, base_position, n_owned, reset=True):
        """One pass over this rank's shard: begin, exchange, finish."""
        self.begin(d_iq, nbytes, encoding, base_position, n_owned, reset=reset)
        self.exchange()
        return self.finish()
