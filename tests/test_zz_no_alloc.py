"""GPU tier: sora_adsb_decode_noalloc() -- scan + resolve + frames to a device
buffer with every list in the library's static device array -- against the
existing entry points it is a variant of (sora_adsb_submit_device +
sora_adsb_collect through a context), bit for bit, and against the oracle.
Collected last: a fault in it must not hide the other tests."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import sora_amd
from conftest import assert_same_frames, frames_as_dicts
from sora_amd import (EMIT_ACCEPTED, EMIT_CANDIDATES, ENC_SC16, ENC_UC8, FRAME_DTYPE,
                      AdsbContext, synth)
from sora_amd._lib import BYTES_PER_SAMPLE, NOALLOC_MAX_SAMPLES, lib

pytestmark = pytest.mark.gpu


def noalloc(d_iq, nbytes, enc, d_frames, d_status, base=0, n_owned=None, emit=EMIT_ACCEPTED,
            options=0, keep_known=False, stream=None):
    """One call through the C ABI; returns (frames, flags) after a device sync."""
    import torch
    n = nbytes // BYTES_PER_SAMPLE[enc]
    cap = d_frames.numel() // FRAME_DTYPE.itemsize
    rc = lib().sora_adsb_decode_noalloc(0, stream, d_iq.data_ptr(), nbytes, enc, base,
                                        n if n_owned is None else n_owned, emit, options,
                                        int(keep_known), d_frames.data_ptr(), cap,
                                        d_status.data_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    total, flags = (int(x) for x in d_status.cpu().numpy().view(np.uint64))
    frames = d_frames.cpu().numpy().view(FRAME_DTYPE)[:min(total, cap)].copy()
    return frames, total, flags


def by_context(raw, enc, emit=EMIT_ACCEPTED, **kw):
    n = len(raw) // BYTES_PER_SAMPLE[enc]
    with AdsbContext(max(n, 1), emit_mode=emit, max_candidates=n // 2 + 1024,
                     max_squitters=n // 2 + 1024, **kw) as ctx:
        buf = sora_amd.DeviceBuffer.from_host(raw + bytes(16))
        ctx.submit_device(buf, len(raw), enc)
        return ctx.collect()


def traffic(n, seed, enc=ENC_UC8):
    p = synth.default_params(4 if enc == ENC_SC16 else 2, n_samples=n, mean_gap=700,
                             reply_permille=300, unknown_permille=250, n_icao=7, seed=seed)
    p.encoding = enc
    return synth.fill_host(n, enc, p.seed, synth.make_frames(p)).tobytes()


def test_noalloc_equals_the_context_path_twice_in_a_row():
    """Two calls in a row on the shared array, the second with a smaller input:
    nothing of the first call may show in the second."""
    import torch
    d_frames = torch.zeros(400000 * FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_status = torch.zeros(2, dtype=torch.int64, device="cuda")
    for n, seed in ((1900000, 61), (333333, 62)):
        raw = traffic(n, seed)
        d_iq = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).cuda()
        for emit in (EMIT_ACCEPTED, EMIT_CANDIDATES):
            want = by_context(raw, ENC_UC8, emit)
            got, total, flags = noalloc(d_iq, len(raw), ENC_UC8, d_frames, d_status, emit=emit)
            assert flags == 0 and total == len(want) > 100
            assert got.tobytes() == want.tobytes()          # all 48 bytes of every frame
        oracle = [c for c in ob.scan(raw) if c["accepted"]]
        got, _, _ = noalloc(d_iq, len(raw), ENC_UC8, d_frames, d_status)
        assert_same_frames(frames_as_dicts(got), oracle)
        assert sum(1 for c in oracle if c["df"] not in (11, 17, 18)) > 10   # the hashed table


def test_noalloc_sc16_correction_and_edges():
    import torch
    d_frames = torch.zeros(300000 * FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_status = torch.zeros(2, dtype=torch.int64, device="cuda")
    raw = traffic(500000, 63, ENC_SC16)
    d_iq = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).cuda()
    want = by_context(raw, ENC_SC16, EMIT_CANDIDATES)
    got, total, flags = noalloc(d_iq, len(raw), ENC_SC16, d_frames, d_status, emit=EMIT_CANDIDATES)
    assert flags == 0 and got.tobytes() == want.tobytes()
    # single-bit correction (parity unpinned) through the options word
    p = synth.default_params(4, n_samples=600000, seed=13)
    p.encoding = ENC_UC8
    p.amp_min, p.amp_max = 30, 110
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    d_iq = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).cuda()
    want = by_context(raw, ENC_UC8, correct_single_bit=True)
    got, _, flags = noalloc(d_iq, len(raw), ENC_UC8, d_frames, d_status,
                            options=sora_amd._lib.OPT_CORRECT_SINGLE_BIT)
    assert flags == 0 and got.tobytes() == want.tobytes()
    assert (got["corrected_bit"] >= 0).sum() > 30
    # empty, shorter than a window, a tile and a bit
    rng = np.random.default_rng(9)
    for n in (0, 1, 239, 240, 241, 7808 + 239, 7808 + 240, 2 * 7808 + 500):
        raw = rng.integers(0, 256, size=2 * n, dtype=np.uint8).tobytes()
        d_iq = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).cuda()
        got, total, flags = noalloc(d_iq, len(raw), ENC_UC8, d_frames, d_status,
                                    emit=EMIT_CANDIDATES)
        assert flags == 0
        assert_same_frames(frames_as_dicts(got), ob.scan(raw))


def test_noalloc_chunks_of_a_stream_keep_the_known_set():
    """keep_known: consecutive haloed chunks decode like the whole capture; a
    call without it starts from an empty set again."""
    import torch
    raw = traffic(1200000, 64)
    want = [c for c in ob.scan(raw) if c["accepted"]]
    d_all = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).cuda()
    d_frames = torch.zeros(200000 * FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_status = torch.zeros(2, dtype=torch.int64, device="cuda")
    cuts = [0, 400000, 400008, 800000, 1200000]
    got = []
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        end = b if b == cuts[-1] else b + 239
        f, _, flags = noalloc(d_all[2 * a:], 2 * (end - a), ENC_UC8, d_frames, d_status, base=a,
                              n_owned=b - a, keep_known=i > 0)
        assert flags == 0
        got += frames_as_dicts(f)
    assert_same_frames(got, want)
    # the last chunk alone, known set cleared: replies of planes introduced earlier go
    a = cuts[-2]
    f, _, _ = noalloc(d_all[2 * a:], 2 * (cuts[-1] - a), ENC_UC8, d_frames, d_status, base=a)
    alone = [c for c in ob.scan(raw[2 * a:], base_position=a) if c["accepted"]]
    assert_same_frames(frames_as_dicts(f), alone)
    assert len(alone) < sum(1 for c in want if c["position"] >= a)


def test_noalloc_limits_are_errors_or_flags():
    import torch
    d_status = torch.zeros(2, dtype=torch.int64, device="cuda")
    d_frames = torch.zeros(64 * FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    big = torch.zeros(2 * (NOALLOC_MAX_SAMPLES + 1) + 16, dtype=torch.uint8, device="cuda")
    L = lib()
    call = lambda ptr, nbytes, enc=ENC_UC8, emit=EMIT_ACCEPTED: L.sora_adsb_decode_noalloc(
        0, None, ptr, nbytes, enc, 0, nbytes // 2, emit, 0, 0, d_frames.data_ptr(), 64,
        d_status.data_ptr())
    assert call(big.data_ptr(), 2 * (NOALLOC_MAX_SAMPLES + 1)) == sora_amd._lib.ERR_INVALID
    assert call(big.data_ptr(), 2 * NOALLOC_MAX_SAMPLES) == 0           # the maximum itself
    torch.cuda.synchronize()
    assert call(big.data_ptr() + 2, 1000) == sora_amd._lib.ERR_ALIGN
    assert call(big.data_ptr(), 1000, enc=7) == sora_amd._lib.ERR_INVALID
    assert call(big.data_ptr(), 1000, emit=5) == sora_amd._lib.ERR_INVALID
    # more frames than the caller's buffer: flagged, the first cap_frames are right
    raw = traffic(900000, 65)
    d_iq = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).cuda()
    want = by_context(raw, ENC_UC8)
    got, total, flags = noalloc(d_iq, len(raw), ENC_UC8, d_frames, d_status)
    assert total == len(want) > 64 and flags == 0x04
    assert got.tobytes() == want[:64].tobytes()
    # the physical bound of the record list: a candidate at every other position
    # cannot exist, one per ten can (dense_candidates of test_gpu_parity)
    n = NOALLOC_MAX_SAMPLES
    v = np.array([9, 1, 8, 1, 2, 2, 2, 5, 1, 3], dtype=np.int64)
    iq = np.full(2 * n, 128, dtype=np.uint8)
    iq[0::2] = 128 + 10 * v[np.arange(n) % 10]
    d_iq = torch.from_numpy(np.concatenate([iq, np.zeros(16, np.uint8)])).cuda()
    d_many = torch.zeros(250000 * FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    got, total, flags = noalloc(d_iq, 2 * n, ENC_UC8, d_many, d_status, emit=EMIT_CANDIDATES)
    assert flags == 0 and total == len(got) == (n - 239 + 9) // 10
    assert np.array_equal(got["position"], np.arange(total, dtype=np.uint64) * 10)


# ---- the header's claims beyond parity: it allocates nothing, leaves the
# caller's current device alone, and can be captured into a HIP graph and replayed
def test_noalloc_allocates_nothing_and_replays_from_a_graph():
    import torch
    raw_a, raw_b = traffic(800000, 71), traffic(800000, 72)
    want_a, want_b = by_context(raw_a, ENC_UC8), by_context(raw_b, ENC_UC8)
    assert len(want_a) > 100 and want_a.tobytes() != want_b.tobytes()
    d_iq = torch.frombuffer(bytearray(raw_a + bytes(16)), dtype=torch.uint8).cuda()
    d_b = torch.frombuffer(bytearray(raw_b + bytes(16)), dtype=torch.uint8).cuda()
    d_frames = torch.zeros(100000 * FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_status = torch.zeros(2, dtype=torch.int64, device="cuda")
    # the first call on the device looks the array up: outside any capture
    got, _, flags = noalloc(d_iq, len(raw_a), ENC_UC8, d_frames, d_status)
    assert flags == 0 and got.tobytes() == want_a.tobytes()
    assert torch.cuda.current_device() == 0
    # free device memory as the driver sees it: unchanged by further calls
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        noalloc(d_iq, len(raw_a), ENC_UC8, d_frames, d_status)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0
    # captured into a graph on a side stream, replayed twice with different input
    # in the same buffer: the replay reads what the buffer holds then
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    d_frames.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        rc = lib().sora_adsb_decode_noalloc(0, torch.cuda.current_stream().cuda_stream,
                                            d_iq.data_ptr(), len(raw_a), ENC_UC8, 0,
                                            len(raw_a) // 2, EMIT_ACCEPTED, 0, 0,
                                            d_frames.data_ptr(), 100000, d_status.data_ptr())
        assert rc == 0
    torch.cuda.synchronize()
    assert int(d_status.cpu()[0]) != len(want_b)         # the capture ran nothing yet
    for raw, want in ((raw_b, want_b), (raw_a, want_a)):
        d_iq.copy_(torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8))
        g.replay()
        torch.cuda.synchronize()
        total, flags = (int(x) for x in d_status.cpu().numpy().view(np.uint64))
        frames = d_frames.cpu().numpy().view(FRAME_DTYPE)[:total]
        assert flags == 0 and frames.tobytes() == want.tobytes()
