"""Time-sharded decoding over several GPUs.

One process per GPU; rank r owns the positions [r*per, (r+1)*per) of the
capture and holds 239 halo samples behind them.  The scan needs no
communication.  The path's one exchange: every rank's clean-squitter records
(position << 24 | address, 8 bytes each) are all-gathered so that each rank's
first-seen table knows the addresses earlier ranks introduced; the compacted
frame lists are then gathered to rank 0, whose concatenation in rank order is
the position order.  `torch.distributed` carries both (backend nccl = RCCL
over xGMI on the GPUs, gloo in the CPU tests); sample data never moves.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import BYTES_PER_SAMPLE, FRAME_DTYPE, HALO

FRAME_BYTES = FRAME_DTYPE.itemsize


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


class ShardedDecoder:
    """Drives one context per rank through scan -> exchange -> resolve -> gather."""

    def __init__(self, ctx, rank, world, device, max_frames, group=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.device, self.max_frames, self.group = device, int(max_frames), group
        self.d_sq = torch.zeros(self.max_frames, dtype=torch.int64, device=device)
        self.d_sq_all = torch.zeros(world * self.max_frames, dtype=torch.int64,
                                    device=device)
        self.d_out = torch.zeros(self.max_frames * FRAME_BYTES, dtype=torch.uint8,
                                 device=device)
        self.d_gather = ([torch.zeros_like(self.d_out) for _ in range(world)]
                         if rank == 0 else None)
        self.counts = torch.zeros(world, dtype=torch.int64, device=device)
        pin = device.type == "cuda"
        n_host = self.max_frames * (world if rank == 0 else 1)
        self.h_pin = torch.empty(n_host * FRAME_BYTES, dtype=torch.uint8, pin_memory=pin)
        self.h_frames = self.h_pin.numpy().view(FRAME_DTYPE)

    def _all_counts(self, k):
        mine = torch.tensor([k], dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(self.counts, mine, group=self.group)
        return [int(x) for x in self.counts.tolist()]

    def step(self, d_iq, nbytes, encoding, base_position, n_owned, reset=True):
        """One pass over this rank's shard.  Returns the total number of frames
        (all ranks); on rank 0, self.h_frames[:n] holds them in position order."""
        ctx, world, rank = self.ctx, self.world, self.rank
        if reset:
            ctx.reset()
        ctx.scan(d_iq, nbytes, encoding, base_position=base_position, n_owned=n_owned)

        ks = self._all_counts(ctx.squitter_count())
        if max(ks) > self.max_frames:
            raise RuntimeError(f"{max(ks)} squitter records exceed max_frames")
        m = max(ks)
        if m:
            ctx.copy_squitters(self.d_sq, self.max_frames)
            ctx.synchronize()
            dist.all_gather_into_tensor(self.d_sq_all[:world * m], self.d_sq[:m],
                                        group=self.group)
            for r in range(world):
                if r != rank and ks[r]:
                    ctx.add_known(self.d_sq_all[r * m:r * m + ks[r]], ks[r])
        ctx.resolve()

        ns = self._all_counts(ctx.collect_device(self.d_out, self.max_frames))
        if max(ns) > self.max_frames:
            raise RuntimeError(f"{max(ns)} frames exceed max_frames")
        m = max(ns) * FRAME_BYTES
        if m:
            dist.gather(self.d_out[:m],
                        [g[:m] for g in self.d_gather] if rank == 0 else None,
                        dst=0, group=self.group)
        if rank == 0:
            off = 0
            for r in range(world):
                nb = ns[r] * FRAME_BYTES
                self.h_pin[off:off + nb].copy_(self.d_gather[r][:nb], non_blocking=True)
                off += nb
            if self.device.type == "cuda":
                torch.cuda.synchronize()
        return sum(ns)
