"""GPU tier (-m gpu): the gfx950 path, called through the C ABI, against the
oracle on the same inputs, against the golden vectors of the compiled
reference, and -- at sizes the oracle cannot walk -- through size-independent
properties.  Bar: bit-exact (positions, 112 bits, amplitude_56 as a double,
DF, syndrome, address, accepted flag)."""
import numpy as np
import pytest

import oracle_binding as ob
import sora_amd
from conftest import GOLDEN_NAMES, assert_same_frames, frames_as_dicts, load_golden
from sora_amd import (EMIT_ACCEPTED, EMIT_CANDIDATES, ENC_SC8, ENC_SC16, ENC_UC8,
                      AdsbContext, DeviceBuffer, synth)

pytestmark = pytest.mark.gpu


def gpu_frames(raw, encoding=ENC_UC8, emit=EMIT_CANDIDATES, host=False, **kw):
    raw = bytes(raw)
    n = len(raw) // sora_amd._lib.BYTES_PER_SAMPLE[encoding]
    with AdsbContext(max(n, 1), emit_mode=emit, max_candidates=n // 2 + 1024,
                     max_squitters=n // 2 + 1024, **kw) as ctx:
        if host:
            ctx.submit_host(raw, encoding)
        else:
            buf = DeviceBuffer.from_host(raw + bytes(16))
            ctx.submit_device(buf, len(raw), encoding)
        frames = ctx.collect()
        counts = ctx.counts()
    return frames_as_dicts(frames), counts


def dense_candidates(n):
    """A period-10 magnitude pattern with a preamble hit at every tenth position:
    1613 candidates per 16128-position tile, more than the scan kernel's small
    hit queue (1536) holds."""
    v = np.array([9, 1, 8, 1, 2, 2, 2, 5, 1, 3], dtype=np.int64)
    iq = np.full(2 * n, 128, dtype=np.uint8)
    iq[0::2] = 128 + 10 * v[np.arange(n) % 10]
    return iq.tobytes()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_golden_candidates(name):
    raw, cands, _ = load_golden(name)
    got, counts = gpu_frames(raw)
    assert_same_frames(got, cands, fields=("position", "bits", "amplitude_56",
                                            "amplitude", "accepted"))
    assert counts["candidates"] == len(cands)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_golden_accepted(name):
    raw, cands, _ = load_golden(name)
    got, _ = gpu_frames(raw, emit=EMIT_ACCEPTED, host=True)
    assert_same_frames(got, [c for c in cands if c["accepted"]],
                       fields=("position", "bits", "amplitude_56", "amplitude",
                               "accepted"))


@pytest.mark.parametrize("enc", [ENC_UC8, ENC_SC8])
def test_uniform_noise_all_fields(enc):
    rng = np.random.default_rng(100 + enc)
    raw = rng.integers(0, 256, size=2 * 300007, dtype=np.uint8).tobytes()
    got, counts = gpu_frames(raw, enc)
    want = ob.scan(raw, enc)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want)
    assert counts["replies"] + counts["squitters"] + counts["dirty_squitters"] == len(want)


@pytest.mark.parametrize("seed", [1, 2])
def test_mixed_traffic_all_fields(seed):
    p = synth.default_params(2, n_samples=1500000, mean_gap=800, reply_permille=200,
                             unknown_permille=300, n_icao=7, seed=seed)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw)
    got, _ = gpu_frames(raw)
    assert_same_frames(got, want)
    acc, _ = gpu_frames(raw, emit=EMIT_ACCEPTED)
    assert_same_frames(acc, [c for c in want if c["accepted"]])
    assert sum(1 for c in want if c["accepted"] and c["df"] not in (11, 17, 18)) > 10


@pytest.mark.parametrize("n", [0, 1, 17, 239, 240, 241, 255, 16128 + 239, 16128 + 240,
                               16367, 16368, 16384, 2 * 16128 + 238,
                               # the default kernel's 7808-position tiles (8064 samples read)
                               7807, 7808, 7808 + 238, 7808 + 239, 7808 + 240, 7808 + 241,
                               8063, 8064, 8065, 2 * 7808 + 238, 2 * 7808 + 239,
                               2 * 7808 + 240, 3 * 7808 - 1 + 240])
def test_edge_lengths(n):
    rng = np.random.default_rng(n)
    raw = rng.integers(0, 256, size=2 * n, dtype=np.uint8).tobytes()
    got, _ = gpu_frames(raw)
    assert_same_frames(got, ob.scan(raw))


def test_ties_and_constant_input():
    """Constant input: every compare is a tie, the strict predicate never fires."""
    got, counts = gpu_frames(bytes([128]) * 100000)
    assert got == [] and counts["candidates"] == 0
    got, _ = gpu_frames(bytes([0, 255] * 50000))
    assert got == []


def test_frames_on_tile_edges():
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    T = 16128
    edges = [T - 241, T + 10, 2 * T - 17, 3 * T - 1, 4 * T, 5 * T + 1, 6 * T - 64,
             6 * T + 256, 7 * T - 239]
    table = synth.frame_table([(p, klm, 70, 0, 112) for p in edges])
    raw = synth.fill_host(7 * T + 500, ENC_UC8, 99, table).tobytes()
    want = ob.scan(raw)
    got, _ = gpu_frames(raw)
    assert_same_frames(got, want)
    assert set(edges) <= {c["position"] for c in got if c["accepted"]}


def test_magnitude_bit_exact():
    """Row a1; north_star allows 1 ulp, the kernels are exact."""
    import torch
    rng = np.random.default_rng(3)
    for enc, nbytes in ((ENC_UC8, 2 * 70001), (ENC_SC8, 2 * 70001), (ENC_SC16, 4 * 70001)):
        raw = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        d = torch.from_numpy(raw).cuda()
        n = nbytes // sora_amd._lib.BYTES_PER_SAMPLE[enc]
        m = torch.empty(n, dtype=torch.float32, device="cuda")
        with AdsbContext(n) as ctx:
            ctx.magnitude(d, nbytes, enc, m)
            ctx.synchronize()
        want = ob.magnitude(raw.tobytes(), enc)
        assert np.array_equal(m.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_streaming_chunks_equal_whole():
    """Known addresses persist across submits (planes never expire, plane.c:90-105);
    a chunk carries a 239-sample halo."""
    p = synth.default_params(2, n_samples=900000, mean_gap=900, reply_permille=250,
                             unknown_permille=200, n_icao=5, seed=77)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    cuts = [0, 250001, 250002, 600000, 900000]
    got = []
    with AdsbContext(400000 + 239) as ctx:
        for a, b in zip(cuts[:-1], cuts[1:]):
            last = b == cuts[-1]
            end = b if last else b + 239
            ctx.submit_host(raw[2 * a:2 * end], ENC_UC8, base_position=a,
                            n_owned=b - a)
            got += frames_as_dicts(ctx.collect())
    assert_same_frames(got, want)


def test_shard_exchange_equals_whole():
    """Two time shards scanned independently; squitter records exchanged."""
    import torch
    p = synth.default_params(2, n_samples=800000, mean_gap=1000, reply_permille=300,
                             unknown_permille=200, n_icao=4, seed=5)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    half = 400000
    shards = [(0, half, half + 239), (half, p.n_samples - half, p.n_samples - half)]
    ctxs = [AdsbContext(half + 239) for _ in shards]
    bufs, recs = [], []
    for ctx, (a, owned, n) in zip(ctxs, shards):
        buf = DeviceBuffer.from_host(raw[2 * a:2 * (a + n)] + bytes(16))
        bufs.append(buf)
        ctx.scan(buf, 2 * n, ENC_UC8, base_position=a, n_owned=owned)
        k = ctx.squitter_count()
        t = torch.zeros(max(k, 1), dtype=torch.int64, device="cuda")
        ctx.copy_squitters(t, k)
        ctx.synchronize()
        recs.append((t, k))
    got = []
    for i, ctx in enumerate(ctxs):
        for j, (t, k) in enumerate(recs):
            if j != i:
                ctx.add_known(t, k)
        ctx.resolve()
        got += frames_as_dicts(ctx.collect())
        ctx.close()
    assert_same_frames(got, want)


@pytest.mark.parametrize("block", [64, 3])
def test_shard_exchange_blocks_equal_whole(block):
    """The same exchange with no host synchronisation: export_block writes
    (count, records) on the device, the blocks of all shards lie end to end as an
    all-gather leaves them, add_known_blocks reads the counts on the device.  A
    block too small for a shard's addresses shows in its count."""
    import torch
    p = synth.default_params(2, n_samples=800000, mean_gap=1000, reply_permille=300,
                             unknown_permille=200, n_icao=4, seed=5)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    half = 400000
    shards = [(0, half, half + 239), (half, p.n_samples - half, p.n_samples - half)]
    ctxs = [AdsbContext(half + 239) for _ in shards]
    blocks = torch.full((2 * block,), -1, dtype=torch.int64, device="cuda")
    bufs = []
    for i, (ctx, (a, owned, n)) in enumerate(zip(ctxs, shards)):
        bufs.append(DeviceBuffer.from_host(raw[2 * a:2 * (a + n)] + bytes(16)))
        ctx.scan(bufs[-1], 2 * n, ENC_UC8, base_position=a, n_owned=owned)
        ctx.export_block(blocks[i * block:], block)
        ctx.synchronize()
    counts = blocks.view(2, block)[:, 0].tolist()
    assert counts == [ctx.squitter_count() for ctx in ctxs] and all(1 <= k <= 4 for k in counts)
    if block == 3:
        assert max(counts) > block - 1          # what ShardedDecoder.finish() raises on
        for ctx in ctxs:
            ctx.close()
        return
    got = []
    for i, ctx in enumerate(ctxs):
        ctx.add_known_blocks(blocks, 2, block, skip_block=i)
        ctx.resolve()
        got += frames_as_dicts(ctx.collect())
        ctx.close()
    assert_same_frames(got, want)


def test_overflow_is_reported_with_fixed_lists():
    """SORA_OPT_FIXED_LISTS: the lists never change, an overflow is an error."""
    rng = np.random.default_rng(1)
    raw = rng.integers(0, 256, size=2 * 200000, dtype=np.uint8).tobytes()
    with AdsbContext(200000, max_candidates=100, fixed_lists=True) as ctx:
        ctx.submit_host(raw, ENC_UC8)
        with pytest.raises(sora_amd.SoraError) as e:
            ctx.collect()
        assert e.value.code == sora_amd._lib.ERR_OVERFLOW
        assert ctx.counts()["overflow"] and ctx.list_regrows() == 0


@pytest.mark.parametrize("small", ["max_candidates", "max_squitters", "max_frames", "all"])
@pytest.mark.parametrize("emit", [EMIT_ACCEPTED, EMIT_CANDIDATES])
def test_overflow_reruns_the_chunk(small, emit):
    """Default: a list that overflowed is grown and the chunk run again -- the
    reference never drops a candidate (plane-iq.c:141-194).  Each list on its own
    and all three together, far too small; dense traffic so that the squitter
    and frame lists matter."""
    p = synth.default_params(2, n_samples=700000, mean_gap=400, reply_permille=300,
                             unknown_permille=200, n_icao=6, seed=41)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    raw = dense_candidates(60000) + raw              # a burst in the first regions only
    n = len(raw) // 2
    want = ob.scan(raw)
    if emit == EMIT_ACCEPTED:
        want = [c for c in want if c["accepted"]]
    assert sum(1 for c in want if c["accepted"]) > 1000
    sizes = {k: 16 for k in ("max_candidates", "max_squitters", "max_frames")
             if small in (k, "all")}
    with AdsbContext(n, emit_mode=emit, **sizes) as ctx:
        for again in range(2):                       # the second pass fits as it is
            ctx.reset()
            ctx.submit_host(raw, ENC_UC8)
            got = frames_as_dicts(ctx.collect())
            counts = ctx.counts()
            assert_same_frames(got, want)
            assert not counts["overflow"] and counts["frames"] == len(want)
            assert ctx.list_regrows() >= 1
            if again == 0:
                first = ctx.list_regrows()
        assert ctx.list_regrows() == first
    # counts() alone triggers the re-run too (ShardedDecoder asks for the frame
    # count before it collects)
    with AdsbContext(n, emit_mode=emit, **sizes) as ctx:
        buf = DeviceBuffer.from_host(raw + bytes(16))
        ctx.submit_device(buf, len(raw), ENC_UC8)
        assert ctx.counts()["frames"] == len(want)
        assert_same_frames(frames_as_dicts(ctx.collect()), want)


def test_frame_list_regrow_keeps_the_record_list():
    """Only the accepted / frame lists too small, with a record list of tens of
    megabytes: the records stay where the scan wrote them and only the resolve
    runs again on new accepted lists.  (Until round 3 every list was replaced
    and the scan was NOT run again, so that the filter read a record buffer fresh
    from hipMalloc; small buffers came back from the allocator with their old
    contents and hid it.)"""
    import torch
    p = synth.default_params(2, n_samples=80000000, mean_gap=20000, reply_permille=200,
                             unknown_permille=200, n_icao=9, seed=77)
    fr = synth.make_frames(p)
    n = p.n_samples
    d = torch.empty(2 * n + 16, dtype=torch.uint8, device="cuda")
    synth.fill_device(d.data_ptr(), n, p.encoding, p.seed, fr)
    with AdsbContext(n) as ctx:                          # lists that hold everything
        ctx.submit_device(d, 2 * n, ENC_UC8)
        want = ctx.collect()
        assert ctx.list_regrows() == 0
    assert len(want) > 2000
    for emit in (EMIT_ACCEPTED,):
        # 5 M record slots of 8 bytes = 40 MB; 16 frames
        with AdsbContext(n, emit_mode=emit, max_frames=16) as ctx:
            # other allocations in between, so that a freed list is not handed back
            junk = [torch.full((8 << 20,), 0x5a, dtype=torch.uint8, device="cuda")
                    for _ in range(4)]
            ctx.submit_device(d, 2 * n, ENC_UC8)
            got = ctx.collect()
            assert 1 <= ctx.list_regrows() <= 3
            del junk
        assert got.tobytes() == want.tobytes()
    # a window of it against the oracle
    lo, m = 30000000, 2000000
    raw = d[2 * lo:2 * (lo + m)].cpu().numpy().tobytes()
    ps = ob.PlaneSet()
    pos = want["position"].astype(np.int64)
    for f in want[pos < lo]:
        if int(f["flags"]) & sora_amd.FRAME_SQUITTER:
            ps.parse(sora_amd.bits_to_string(f["bits"]))
    ref = [c for c in ob.scan(raw, base_position=lo, plane_set=ps) if c["accepted"]]
    assert_same_frames(frames_as_dicts(want[(pos >= lo) & (pos < lo + m - 239)]), ref)


def test_overflow_reruns_a_streamed_block():
    """The same through sora_adsb_stream_*: the pinned frame buffers follow."""
    p = synth.default_params(2, n_samples=500000, mean_gap=300, reply_permille=300,
                             unknown_permille=200, n_icao=6, seed=43)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = [c for c in ob.scan(raw) if c["accepted"]]
    with AdsbContext(120000 + 239, max_candidates=16, max_squitters=16, max_frames=16) as ctx:
        got = frames_as_dicts(np.concatenate(list(ctx.stream(raw, ENC_UC8, 120000))))
        assert ctx.list_regrows() >= 1      # (a record list is one 256-record unit at least)
    assert_same_frames(got, want)
    want = ob.scan(raw)
    with AdsbContext(120000 + 239, emit_mode=EMIT_CANDIDATES, max_candidates=16,
                     max_squitters=16, max_frames=16) as ctx:
        got = frames_as_dicts(np.concatenate(list(ctx.stream(raw, ENC_UC8, 120000))))
    assert_same_frames(got, want)


def test_overflow_after_an_export_is_reported():
    """A squitter list that overflowed after its export went to other shards
    cannot be repaired locally: they merged a truncated list."""
    import torch
    p = synth.default_params(2, n_samples=400000, mean_gap=400, n_icao=6, seed=44)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    buf = DeviceBuffer.from_host(raw + bytes(16))
    blk = torch.zeros(64, dtype=torch.int64, device="cuda")
    with AdsbContext(p.n_samples, max_squitters=16) as ctx:
        ctx.scan(buf, len(raw), ENC_UC8)
        ctx.export_block(blk, 64)
        ctx.resolve()
        with pytest.raises(sora_amd.SoraError) as e:
            ctx.collect()
        assert e.value.code == sora_amd._lib.ERR_OVERFLOW
    # the record and frame lists are repaired even then: the export was whole
    want = [c for c in ob.scan(raw) if c["accepted"]]
    with AdsbContext(p.n_samples, max_candidates=16, max_frames=16) as ctx:
        ctx.scan(buf, len(raw), ENC_UC8)
        ctx.export_block(blk, 64)
        ctx.resolve()
        assert_same_frames(frames_as_dicts(ctx.collect()), want)
        assert ctx.list_regrows() >= 1


def test_bad_arguments():
    with AdsbContext(1000) as ctx:
        buf = DeviceBuffer(4096)
        with pytest.raises(sora_amd.SoraError) as e:
            ctx.scan(buf.ptr + 2, 100, ENC_UC8)
        assert e.value.code == sora_amd._lib.ERR_ALIGN
        with pytest.raises(sora_amd.SoraError) as e:
            ctx.scan(buf, 4096, ENC_UC8)            # 2048 samples > max_chunk 1000
        assert e.value.code == sora_amd._lib.ERR_INVALID
        with pytest.raises(sora_amd.SoraError) as e:
            ctx.collect()
        assert e.value.code == sora_amd._lib.ERR_STATE


def test_device_fill_equals_host_fill():
    import torch
    p = synth.default_params(2, n_samples=3000001, mean_gap=600)
    fr = synth.make_frames(p)
    for enc in (ENC_UC8, ENC_SC8, ENC_SC16):
        nb = p.n_samples * sora_amd._lib.BYTES_PER_SAMPLE[enc]
        d = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        fr2 = fr.copy()
        if enc == ENC_SC16:
            fr2["amp_i"] *= 100
            fr2["amp_q"] *= 100
        synth.fill_device(d.data_ptr(), p.n_samples, enc, p.seed, fr2)
        want = synth.fill_host(p.n_samples, enc, p.seed, fr2)
        assert np.array_equal(d.cpu().numpy(), want)


def test_config1_size_properties():
    """BASELINE config 1 size (120 M samples, 240 MB): properties that need no
    oracle walk -- sorted output, every strong injected DF17 recovered
    bit-exactly, idempotence, and the chunked result equal to the whole."""
    import torch
    p = synth.default_params(1)
    fr = synth.make_frames(p)
    d = torch.empty(2 * p.n_samples, dtype=torch.uint8, device="cuda")
    synth.fill_device(d.data_ptr(), p.n_samples, p.encoding, p.seed, fr)
    with AdsbContext(p.n_samples) as ctx:
        ctx.submit_device(d, 2 * p.n_samples, ENC_UC8)
        a = ctx.collect()
        ctx.reset()
        ctx.submit_device(d, 2 * p.n_samples, ENC_UC8)
        b = ctx.collect()
        counts = ctx.counts()
    assert a.tobytes() == b.tobytes()                       # idempotent
    pos = a["position"].astype(np.int64)
    assert np.all(np.diff(pos) > 0)                         # sorted, unique
    assert 0.015 < counts["candidates"] / p.n_samples < 0.03
    got = {int(f["position"]): f for f in a}
    df17 = [f for f in fr if int(f["kind"]) == synth.KIND_DF17]
    hit = sum(1 for f in df17 if int(f["position"]) in got and
              np.array_equal(got[int(f["position"])]["bits"], f["bits"]))
    assert hit >= 0.9 * len(df17)
    # a 4 M-sample window in the middle against the oracle, exactly
    lo, n = 60000000, 4000000
    raw = d[2 * lo:2 * (lo + n)].cpu().numpy().tobytes()
    ps = ob.PlaneSet()
    for f in a[pos < lo]:                                   # addresses known so far
        if int(f["flags"]) & sora_amd.FRAME_SQUITTER:
            ps.parse(sora_amd.bits_to_string(f["bits"]))
    want = [c for c in ob.scan(raw, base_position=lo, plane_set=ps) if c["accepted"]]
    window = frames_as_dicts(a[(pos >= lo) & (pos < lo + n - 239)])
    assert_same_frames(window, want)
    # chunked with halos, 7 ragged chunks
    # (a device chunk must start 16-byte aligned: shard starts are multiples of 8)
    cuts = np.linspace(0, p.n_samples, 8).astype(np.int64) // 8 * 8
    cuts[1:-1] += np.array([8, 56, 64, 136, 1000, 16120])
    out = []
    with AdsbContext(int(np.diff(cuts).max()) + 239) as ctx:
        for s, e in zip(cuts[:-1], cuts[1:]):
            end = int(e) if e == cuts[-1] else int(e) + 239
            ctx.submit_device(d.data_ptr() + 2 * int(s), 2 * (end - int(s)), ENC_UC8,
                              base_position=int(s), n_owned=int(e - s))
            out.append(ctx.collect())
    assert np.concatenate(out).tobytes() == a.tobytes()


# ---- sc16: adsb_scan_kernel<true>.  Parity unpinned by
# the reference (it has no sc16 route into ADS-B); the oracle's definition of the
# sc16 magnitude is the build's own, see oracle/adsb_oracle.c.

def sc16_noise(seed, n, sigma):
    rng = np.random.default_rng(seed)
    v = np.clip(np.rint(rng.normal(0, sigma, size=2 * n)), -32768, 32767)
    return v.astype("<i2").tobytes()


@pytest.mark.parametrize("sigma", [3, 400, 8000, 30000])
def test_sc16_noise_all_fields(sigma):
    """sigma 8000 and 30000 put I^2+Q^2 past 2^24 (float rounding ties, clipping)."""
    raw = sc16_noise(40 + sigma, 200003, sigma)
    want = ob.scan(raw, ENC_SC16)
    got, counts = gpu_frames(raw, ENC_SC16)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want) > 1000


def test_sc16_uniform_bytes():
    rng = np.random.default_rng(8)
    raw = rng.integers(0, 256, size=4 * 100001, dtype=np.uint8).tobytes()
    got, _ = gpu_frames(raw, ENC_SC16)
    assert_same_frames(got, ob.scan(raw, ENC_SC16))


@pytest.mark.parametrize("n", [0, 1, 239, 240, 241, 7936 + 239, 7936 + 240, 8191, 8192,
                               2 * 7936 + 238])
def test_sc16_edge_lengths(n):
    raw = sc16_noise(n, n, 400)
    got, _ = gpu_frames(raw, ENC_SC16)
    assert_same_frames(got, ob.scan(raw, ENC_SC16))


def test_sc16_config4_traffic():
    """BASELINE configs[3] in small: dense sc16 traffic, overlapping frames, 20 %
    of them with one flipped bit (rejected: correction is opt-in, D3)."""
    p = synth.default_params(4, n_samples=1200000, seed=5)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw, ENC_SC16)
    got, _ = gpu_frames(raw, ENC_SC16)
    assert_same_frames(got, want)
    acc, _ = gpu_frames(raw, ENC_SC16, emit=EMIT_ACCEPTED, host=True)
    assert_same_frames(acc, [c for c in want if c["accepted"]])
    assert len(acc) > 1000


@pytest.mark.parametrize("emit", [EMIT_ACCEPTED, EMIT_CANDIDATES])
def test_sc16_config4_traffic_with_correction(emit):
    """BASELINE configs[3] as it reads: dense overlapping sc16 traffic, 20 % of the
    frames with one flipped bit, AND the correction path on -- against the oracle
    (parity unpinned by the reference, which never corrects and has no sc16 route:
    the oracle restates this build's definitions), whole and through the streaming
    ingest."""
    p = synth.default_params(4, n_samples=1300000, seed=15, reply_permille=150,
                             unknown_permille=300)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw, ENC_SC16, correct=True)
    assert sum(1 for c in want if c["corrected_bit"] >= 0) > 200
    if emit == EMIT_ACCEPTED:
        want = [c for c in want if c["accepted"]]
    assert sum(1 for c in want if c["accepted"] and c["df"] not in (11, 17, 18)) > 10
    got, counts = gpu_frames(raw, ENC_SC16, emit=emit, correct_single_bit=True)
    assert_same_frames(got, want)
    n = p.n_samples
    with AdsbContext(300000 + 239, emit_mode=emit, max_candidates=n // 2, max_squitters=n // 2,
                     correct_single_bit=True) as ctx:
        parts = list(ctx.stream(raw, ENC_SC16, 300000))
    assert len(parts) == 5
    assert_same_frames(frames_as_dicts(np.concatenate(parts)), want)


def test_sc16_frames_on_tile_edges():
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    T = 7936
    edges = [T - 241, T + 10, 2 * T - 17, 3 * T - 1, 4 * T, 5 * T + 1, 6 * T - 64,
             6 * T + 256, 7 * T - 239]
    table = synth.frame_table([(p, klm, 9000, 0, 112) for p in edges])
    raw = synth.fill_host(7 * T + 500, ENC_SC16, 99, table).tobytes()
    want = ob.scan(raw, ENC_SC16)
    got, _ = gpu_frames(raw, ENC_SC16)
    assert_same_frames(got, want)
    assert set(edges) <= {c["position"] for c in got if c["accepted"]}


# ---- single-bit correction, opt-in; parity unpinned:
# the reference never corrects, the oracle restates this build's definition.

def test_correction_off_by_default_and_on():
    p = synth.default_params(4, n_samples=1500000, seed=9)      # 20 % flipped-bit frames
    p.encoding = ENC_UC8
    p.amp_min, p.amp_max = 30, 110
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    plain = ob.scan(raw)
    fixed = ob.scan(raw, correct=True)
    n_fixed = sum(1 for c in fixed if c["corrected_bit"] >= 0)
    assert n_fixed > 100 and all(c["corrected_bit"] < 0 for c in plain)
    got, counts = gpu_frames(raw)                               # option off
    assert_same_frames(got, plain)
    assert counts["corrected"] == 0
    n = len(raw) // 2
    with AdsbContext(n, emit_mode=EMIT_CANDIDATES, max_candidates=n // 2,
                     max_squitters=n // 2, correct_single_bit=True) as ctx:
        ctx.submit_host(raw, ENC_UC8)
        got = frames_as_dicts(ctx.collect())
        counts = ctx.counts()
    assert_same_frames(got, fixed)
    assert counts["corrected"] == n_fixed
    with AdsbContext(n, max_squitters=n // 2, correct_single_bit=True) as ctx:
        ctx.submit_host(raw, ENC_UC8)
        acc = frames_as_dicts(ctx.collect())
    assert_same_frames(acc, [c for c in fixed if c["accepted"]])


def test_correction_sc16_every_bit():
    """One clean DF17 and one DF11 per flipped bit position: 5..111 and 5..55."""
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    df11 = synth.hex_to_bits("5D4840D6F8740F")
    rows, pos = [], 400
    for b in range(112):
        w = np.array(klm, dtype=np.uint32).copy()
        w[b >> 5] ^= np.uint32(0x80000000 >> (b & 31))
        rows.append((pos, w, 9000, 0, 112))
        pos += 300
    for b in range(56):
        w = np.array(df11, dtype=np.uint32).copy()
        w[b >> 5] ^= np.uint32(0x80000000 >> (b & 31))
        rows.append((pos, w, 9000, 0, 56))
        pos += 300
    raw = synth.fill_host(pos + 600, ENC_SC16, 3, synth.frame_table(rows)).tobytes()
    want = ob.scan(raw, ENC_SC16, correct=True)
    fixed_bits = sorted(c["corrected_bit"] for c in want if c["corrected_bit"] >= 0
                        and c["position"] >= 400 and (c["position"] - 400) % 300 == 0)
    assert fixed_bits == sorted(list(range(5, 112)) + list(range(5, 56)))
    n = len(raw) // 4
    with AdsbContext(n, emit_mode=EMIT_CANDIDATES, max_candidates=n // 2,
                     max_squitters=n // 2, correct_single_bit=True) as ctx:
        ctx.submit_host(raw, ENC_SC16)
        got = frames_as_dicts(ctx.collect())
    assert_same_frames(got, want)


@pytest.mark.parametrize("correct", [False, True])
def test_sc16_more_squitters_in_a_tile_than_the_scan_sums(correct):
    """The sc16 scan kernel sums the amplitudes of up to SORA_AMP_SLOTS (32) clean
    squitters per 7936-position tile itself and leaves the others to finalize
    (sora_scan_args::sq_sum): short squitters 130 samples apart are 61 to a tile,
    over four tiles, every fifth with a flipped bit (a corrected squitter takes a
    slot but is summed by finalize), every amplitude against the oracle's; then
    the same capture through a context whose lists start too small (regrow)."""
    df11 = synth.hex_to_bits("5D4840D6F8740F")
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    rows, pos = [], 300
    for i in range(240):
        w = np.array(df11, dtype=np.uint32).copy()
        if i % 5 == 4:
            b = 5 + (7 * i) % 51
            w[b >> 5] ^= np.uint32(0x80000000 >> (b & 31))
        rows.append((pos, w, 3000 + 37 * (i % 100), 0, 56))
        pos += 130
    for i in range(20):                                  # and a sparse stretch of long ones
        rows.append((pos, np.array(klm, dtype=np.uint32), 5000 + 211 * i, 0, 112))
        pos += 2500
    raw = synth.fill_host(pos + 600, ENC_SC16, 11, synth.frame_table(rows)).tobytes()
    n = len(raw) // 4
    want = ob.scan(raw, ENC_SC16, correct=correct)
    clean = [c for c in want if c["df"] in (11, 17) and c["syndrome"] == 0
             and c["corrected_bit"] < 0]
    per_tile = np.bincount([c["position"] // 7936 for c in clean])
    assert per_tile.max() > 40 and (per_tile > 32).sum() >= 3
    if correct:
        assert sum(c["corrected_bit"] >= 0 for c in want) >= 40
    for emit in (EMIT_CANDIDATES, EMIT_ACCEPTED):
        sel = want if emit == EMIT_CANDIDATES else [c for c in want if c["accepted"]]
        for small in (False, True):
            kw = (dict(max_candidates=2048, max_squitters=64, max_frames=64) if small else
                  dict(max_candidates=n // 2, max_squitters=n // 2))
            with AdsbContext(n, emit_mode=emit, correct_single_bit=correct, **kw) as ctx:
                ctx.submit_host(raw, ENC_SC16)
                got = frames_as_dicts(ctx.collect())
                assert small == (ctx.list_regrows() > 0)
            assert_same_frames(got, sel)


def test_config4_size_properties(monkeypatch):
    """BASELINE configs[3] at 120 M samples (480 MB of sc16, dense overlapping
    traffic, a fifth of the frames with a flipped bit, correction on): sorted
    output, idempotence, the same frames byte for byte whether the scan kernel or
    finalize sums a squitter's amplitude (SORA_ADSB_SCAN_SUMS), a 2 M-sample
    window in the middle against the oracle, ragged haloed chunks equal to the
    whole."""
    import torch
    p = synth.default_params(4)
    fr = synth.make_frames(p)
    n = p.n_samples
    d = torch.empty(4 * n, dtype=torch.uint8, device="cuda")
    synth.fill_device(d.data_ptr(), n, ENC_SC16, p.seed, fr)
    kw = dict(correct_single_bit=True, max_candidates=n // 8, max_squitters=n // 128,
              max_frames=n // 128)
    with AdsbContext(n, **kw) as ctx:
        ctx.submit_device(d, 4 * n, ENC_SC16)
        a = ctx.collect()
        ctx.reset()
        ctx.submit_device(d, 4 * n, ENC_SC16)
        b = ctx.collect()
        assert ctx.scan_kernel_name(ENC_SC16) == "adsb_scan_kernel<true>"
    assert a.tobytes() == b.tobytes()                       # idempotent
    pos = a["position"].astype(np.int64)
    assert np.all(np.diff(pos) > 0) and len(a) > n // 1000
    assert (a["corrected_bit"] >= 0).sum() > len(a) // 20
    monkeypatch.setenv("SORA_ADSB_SCAN_SUMS", "0")
    with AdsbContext(n, **kw) as ctx:
        ctx.submit_device(d, 4 * n, ENC_SC16)
        assert ctx.collect().tobytes() == a.tobytes()       # finalize's sums are the scan's
    monkeypatch.delenv("SORA_ADSB_SCAN_SUMS")
    lo, m = 60000000, 2000000
    raw = d[4 * lo:4 * (lo + m)].cpu().numpy().tobytes()
    ps = ob.PlaneSet()
    for f in a[pos < lo]:                                   # addresses known so far
        if int(f["flags"]) & sora_amd.FRAME_SQUITTER:
            ps.parse(sora_amd.bits_to_string(f["bits"]))
    want = [c for c in ob.scan(raw, ENC_SC16, base_position=lo, plane_set=ps, correct=True)
            if c["accepted"]]
    assert len(want) > 1000
    assert_same_frames(frames_as_dicts(a[(pos >= lo) & (pos < lo + m - 239)]), want)
    cuts = np.linspace(0, n, 6).astype(np.int64) // 8 * 8
    cuts[1:-1] += np.array([8, 64, 1000, 7936])
    out = []
    with AdsbContext(int(np.diff(cuts).max()) + 239, **kw) as ctx:
        for s_, e in zip(cuts[:-1], cuts[1:]):
            end = int(e) if e == cuts[-1] else int(e) + 239
            ctx.submit_device(d.data_ptr() + 4 * int(s_), 4 * (end - int(s_)), ENC_SC16,
                              base_position=int(s_), n_owned=int(e - s_))
            out.append(ctx.collect())
    assert np.concatenate(out).tobytes() == a.tobytes()


# ---- streaming ingest: sora_adsb_stream_*

@pytest.mark.parametrize("enc,block", [(ENC_UC8, 1024), (ENC_UC8, 50000), (ENC_UC8, 400000),
                                       (ENC_SC16, 30000)])
def test_stream_blocks_equal_whole(enc, block):
    """Blocks through pinned double buffers with a 239-sample carry give the
    frames of the whole capture; a block boundary that is also the end of the
    stream and a trailing partial sample are covered by the sizes."""
    p = synth.default_params(4 if enc == ENC_SC16 else 2, n_samples=400000, mean_gap=700,
                             reply_permille=250, unknown_permille=200, n_icao=5, seed=31)
    p.encoding = enc
    raw = synth.fill_host(p.n_samples, enc, p.seed, synth.make_frames(p)).tobytes() + b"\x07"
    want = [c for c in ob.scan(raw, enc) if c["accepted"]]
    assert len(want) > 50
    n = 400000
    with AdsbContext(block + 239, max_squitters=block + 1024) as ctx:
        parts = list(ctx.stream(raw, enc, block))
    assert len(parts) == n // block + 1
    got = frames_as_dicts(np.concatenate(parts))
    assert_same_frames(got, want)


def test_stream_empty_and_errors():
    with AdsbContext(100000) as ctx:
        assert [len(x) for x in ctx.stream(b"", ENC_UC8, 4096)] == [0]
        with pytest.raises(sora_amd.SoraError):
            list(ctx.stream(b"", ENC_UC8, 100))            # block too small
        with pytest.raises(sora_amd.SoraError):
            list(ctx.stream(b"", ENC_UC8, 100000))         # block + 239 > max_chunk
        # a file object as the source
        import io
        rng = np.random.default_rng(2)
        raw = rng.integers(0, 256, size=2 * 30000, dtype=np.uint8).tobytes()
        got = frames_as_dicts(np.concatenate(list(ctx.stream(io.BytesIO(raw), ENC_UC8, 8192))))
        assert_same_frames(got, [c for c in ob.scan(raw) if c["accepted"]])


@pytest.mark.parametrize("enc", [ENC_UC8, ENC_SC8, ENC_SC16])
def test_wire_frames_unpack_to_the_same_frames(enc):
    """SORA_OPT_WIRE_FRAMES: 28 bytes per frame over PCIe; sora_adsb_unpack_frames
    gives back the 48-byte frame, field for field (every byte), for every
    candidate of mixed traffic, with and without single-bit correction."""
    p = synth.default_params(4 if enc == ENC_SC16 else 2, n_samples=700000, mean_gap=500,
                             reply_permille=300, unknown_permille=200, n_icao=6, seed=71)
    p.encoding = enc
    if enc != ENC_SC16:
        p.flip_permille = 200
        p.amp_min, p.amp_max = 30, 110
    raw = synth.fill_host(p.n_samples, enc, p.seed, synth.make_frames(p)).tobytes()
    n = p.n_samples
    for emit, fix in ((EMIT_CANDIDATES, True), (EMIT_ACCEPTED, False)):
        kw = dict(emit_mode=emit, max_candidates=n // 2, max_squitters=n // 2,
                  correct_single_bit=fix)
        with AdsbContext(n, **kw) as ctx:
            ctx.submit_host(raw, enc)
            full = ctx.collect()
            with pytest.raises(sora_amd.SoraError):
                ctx._collect_fn = lambda device=False: sora_amd._lib.lib().sora_adsb_collect_wire
                ctx.collect()                       # not a wire context
        with AdsbContext(n, wire_frames=True, **kw) as ctx:
            ctx.submit_host(raw, enc)
            wire = ctx.collect()
            assert wire.dtype.itemsize == 28
            with pytest.raises(sora_amd.SoraError):
                list(ctx.stream(raw, enc, 100000))  # the stream calls hand out full frames
        back = sora_amd.unpack_frames(wire, enc)
        assert len(full) > 500 and back.tobytes() == full.tobytes()
        if fix:
            assert (full["corrected_bit"] >= 0).sum() > 20
    # positions beyond 2^32 and the largest amplitudes survive the packing
    raw = sc16_noise(5, 20000, 30000) if enc == ENC_SC16 else bytes([0, 0, 255, 255] * 10000)
    base = (1 << 39) + 123456
    for wire_mode in (False, True):
        with AdsbContext(20000, emit_mode=EMIT_CANDIDATES, wire_frames=wire_mode) as ctx:
            ctx.submit_host(raw, enc, base_position=base)
            got = ctx.collect()
        if wire_mode:
            assert sora_amd.unpack_frames(got, enc).tobytes() == ref.tobytes()
        else:
            ref = got
    if enc == ENC_SC16:
        assert len(ref) > 100 and ref["amplitude_56"].max() > 56 * 0.5


@pytest.mark.parametrize("wire", [False, True])
def test_collect_behind_collects_of_other_sizes(wire):
    """A collect to the host sends as many frames as the context's last one found
    (and a few more) on their way before it knows the count: captures with more
    frames than the one before, with fewer, with none, a second collect of the
    same capture and a destination smaller than the list all hand out what a
    fresh context does."""
    rng = np.random.default_rng(77)
    sizes = [200000, 450000, 460000, 3000, 900000, 100000]
    raws = [rng.integers(0, 256, size=2 * n, dtype=np.uint8).tobytes() for n in sizes]
    raws[3] = bytes(2 * sizes[3])                           # silence: no frame at all
    kw = dict(emit_mode=EMIT_CANDIDATES, wire_frames=wire)
    fresh = []
    for raw, n in zip(raws, sizes):
        with AdsbContext(n, max_candidates=n // 2 + 1024, **kw) as ctx:
            ctx.submit_host(raw, ENC_UC8)
            fresh.append(ctx.collect())
    assert len(fresh[3]) == 0 and len(fresh[4]) > 1.9 * len(fresh[1]) > 3.8 * len(fresh[0])
    n = max(sizes)
    with AdsbContext(n, max_candidates=n // 2 + 1024, **kw) as ctx:
        for raw, want in zip(raws, fresh):
            ctx.reset()
            ctx.submit_host(raw, ENC_UC8)
            assert ctx.collect().tobytes() == want.tobytes()
            assert ctx.collect().tobytes() == want.tobytes()
            few = np.zeros(min(50, len(want)), dtype=want.dtype)
            assert ctx.collect_into(few) == len(want)
            assert few.tobytes() == want[:len(few)].tobytes()
            assert ctx.collect().tobytes() == want.tobytes()


def test_stream_source_failure_is_not_end_of_stream():
    """An exception inside the read callback (ctypes would swallow it and the
    library would see a short read = end of stream) reaches the caller."""
    import io

    class Broken(io.RawIOBase):
        def __init__(self, data, fail_at):
            self.inner, self.fail_at, self.count = io.BytesIO(data), fail_at, 0

        def readinto(self, b):
            self.count += 1
            if self.count > self.fail_at:
                raise OSError("disk on fire")
            return self.inner.readinto(b)

    raw = np.random.default_rng(5).integers(0, 256, size=2 * 60000, dtype=np.uint8).tobytes()
    with AdsbContext(8192 + 239) as ctx:
        with pytest.raises(OSError, match="disk on fire"):
            list(ctx.stream(Broken(raw, 3), ENC_UC8, 8192))
        # and the context is usable afterwards
        got = frames_as_dicts(np.concatenate(list(ctx.stream(Broken(raw, 1000), ENC_UC8, 8192))))
    assert_same_frames(got, [c for c in ob.scan(raw) if c["accepted"]])


def test_config5_batched_archive():
    """BASELINE configs[4] in small: independent recordings (uc8 and sc8) through
    one context, sora_adsb_reset() between them; plane sets must not leak."""
    recs = []
    for i in range(6):
        enc = ENC_UC8 if i % 2 == 0 else ENC_SC8
        p = synth.default_params(1, n_samples=250000, mean_gap=900, reply_permille=300,
                                 unknown_permille=0, n_icao=4, seed=500 + i)
        p.encoding = enc
        recs.append((enc, synth.fill_host(p.n_samples, enc, p.seed,
                                          synth.make_frames(p)).tobytes()))
    with AdsbContext(250000, max_squitters=100000) as ctx:
        for enc, raw in recs:
            ctx.reset()
            ctx.submit_host(raw, enc)
            got = frames_as_dicts(ctx.collect())
            assert_same_frames(got, [c for c in ob.scan(raw, enc) if c["accepted"]])
        # without the reset the planes of the previous recording stay known
        # (plane.c:90-105 never expires them): more replies are accepted
        enc, raw = recs[0]
        ctx.reset()
        ctx.submit_host(recs[2][1], recs[2][0])
        ctx.collect()
        ctx.submit_host(raw, enc)
        leaked = ctx.collect()
        ps = ob.PlaneSet()
        ob.scan(recs[2][1], recs[2][0], plane_set=ps)
        want = [c for c in ob.scan(raw, enc, plane_set=ps) if c["accepted"]]
        assert_same_frames(frames_as_dicts(leaked), want)


def test_config2_full_size_properties():
    """BASELINE configs[1] at its full size (7.2 G samples, 14.4 GB; positions
    pass 2^32): idempotence, order, recovery of the strong injected frames, a
    window behind position 2^32 against the oracle exactly, and the capture as
    two time shards with the squitter exchange (configs[2] with two ranks on one
    device) equal to the whole."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free HBM")
    p = synth.default_params(2)
    assert p.n_samples == 7200000000
    fr = synth.make_frames(p)
    n = p.n_samples
    d = torch.empty(2 * n + 16, dtype=torch.uint8, device="cuda")
    synth.fill_device(d.data_ptr(), n, p.encoding, p.seed, fr)
    cap = n // 1024 + 65536
    with AdsbContext(n, max_squitters=cap, max_frames=cap) as ctx:
        ctx.submit_device(d, 2 * n, ENC_UC8)
        a = ctx.collect()
        counts = ctx.counts()
        ctx.reset()
        ctx.submit_device(d, 2 * n, ENC_UC8)
        b = ctx.collect()
    assert a.tobytes() == b.tobytes()
    pos = a["position"].astype(np.int64)
    assert np.all(np.diff(pos) > 0) and pos[-1] > (1 << 32)
    assert counts["frames"] == len(a)
    assert counts["candidates"] == counts["replies"] + counts["squitters"] + \
        counts["dirty_squitters"]
    # strong frames (amplitude >= 40 counts) are found where they were put
    got = dict(zip(pos.tolist(), range(len(a))))
    strong = [f for f in fr if int(f["kind"]) in (synth.KIND_DF17, synth.KIND_DF11)
              and int(f["amp_i"]) >= 40][::50]
    def same(f):                            # a short frame's bits 56.. are noise
        g = a[got[int(f["position"])]]["bits"]
        w = 4 if int(f["nbits"]) == 112 else 1
        return np.array_equal(g[:w], f["bits"][:w])
    hit = sum(1 for f in strong if int(f["position"]) in got and same(f))
    assert hit >= 0.9 * len(strong)
    # 3 M samples behind 2^32 against the oracle
    lo, m = (1 << 32) + 123456, 3000000
    raw = d[2 * lo:2 * (lo + m)].cpu().numpy().tobytes()
    ps = ob.PlaneSet()
    for f in a[pos < lo]:
        if int(f["flags"]) & sora_amd.FRAME_SQUITTER:
            ps.parse(sora_amd.bits_to_string(f["bits"]))
    want = [c for c in ob.scan(raw, base_position=lo, plane_set=ps) if c["accepted"]]
    assert len(want) > 300
    assert_same_frames(frames_as_dicts(a[(pos >= lo) & (pos < lo + m - 239)]), want)
    # two shards, squitter records exchanged
    half = n // 2
    shards = [(0, half, half + 239), (half, n - half, n - half)]
    ctxs = [AdsbContext(half + 239, max_squitters=cap, max_frames=cap) for _ in shards]
    recs = []
    for ctx, (s, owned, k) in zip(ctxs, shards):
        ctx.scan(d.data_ptr() + 2 * s, 2 * k, ENC_UC8, base_position=s, n_owned=owned)
        c = ctx.squitter_count()
        t = torch.zeros(max(c, 1), dtype=torch.int64, device="cuda")
        ctx.copy_squitters(t, c)
        ctx.synchronize()
        recs.append((t, c))
    parts = []
    for i, ctx in enumerate(ctxs):
        for j, (t, c) in enumerate(recs):
            if j != i:
                ctx.add_known(t, c)
        ctx.resolve()
        parts.append(ctx.collect())
        ctx.close()
    assert np.concatenate(parts).tobytes() == a.tobytes()


def test_more_hits_than_the_small_queue():
    raw = dense_candidates(3 * 16128 + 1000)
    want = ob.scan(raw)
    assert len(want) > 3 * 1600
    got, counts = gpu_frames(raw)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want)
    # and mixed with a sparse tail: tiles switch between the two queues
    raw2 = raw + np.random.default_rng(4).integers(0, 256, 2 * 40000, dtype=np.uint8).tobytes()
    got, _ = gpu_frames(raw2)
    assert_same_frames(got, ob.scan(raw2))


def densest_candidates(n):
    """A period-7 pattern, a preamble hit at every seventh position (the predicate
    looks six positions ahead of a hit for something smaller): 1115 candidates per
    7808-position tile of the default scan kernel, more than its hit queue (1024)
    and more than two flush units of its record ring hold."""
    v = np.array([9, 1, 8, 1, 2, 2, 2], dtype=np.int64)
    iq = np.full(2 * n, 128, dtype=np.uint8)
    iq[0::2] = 128 + 10 * v[np.arange(n) % 7]
    return iq.tobytes()


def test_more_hits_than_the_hit_queue_and_the_record_ring():
    """Tiles whose candidates are worked off in two windows of the hit queue, the
    records leaving through the LDS ring in between, next to sparse tiles; with a
    record list that holds them and with one that is grown (regions that overflow
    in the middle of a flush unit)."""
    raw = densest_candidates(5 * 7808 + 500)
    want = ob.scan(raw)
    assert len(want) > 5 * 1100
    got, counts = gpu_frames(raw)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want)
    rng = np.random.default_rng(14)
    raw2 = (rng.integers(0, 256, 2 * 30000, dtype=np.uint8).tobytes() + raw +
            rng.integers(0, 256, 2 * 50000, dtype=np.uint8).tobytes() +
            dense_candidates(20000))
    want2 = ob.scan(raw2)
    got, _ = gpu_frames(raw2)
    assert_same_frames(got, want2)
    n = len(raw2) // 2
    with AdsbContext(n, emit_mode=EMIT_CANDIDATES, max_candidates=4096, max_frames=4096) as ctx:
        ctx.submit_host(raw2, ENC_UC8)
        frames = frames_as_dicts(ctx.collect())
        assert ctx.list_regrows() >= 1
    assert_same_frames(frames, want2)


@pytest.mark.parametrize("min_tiles,rounds", [(1, 16), (3, 1), (8, 2)])
def test_workgroup_ranges_do_not_matter(monkeypatch, min_tiles, rounds):
    """The scan hands tile ranges to workgroups (at least SORA_ADSB_SCAN_MIN_TILES
    tiles each, SORA_ADSB_SCAN_ROUNDS rounds of workgroups per CU); the record
    regions of the ranges are concatenated.  One tile per workgroup puts every
    tile edge on a region edge; the result is the same list."""
    monkeypatch.setenv("SORA_ADSB_SCAN_MIN_TILES", str(min_tiles))
    monkeypatch.setenv("SORA_ADSB_SCAN_ROUNDS", str(rounds))
    p = synth.default_params(2, n_samples=1200000, mean_gap=700, reply_permille=200,
                             unknown_permille=300, n_icao=7, seed=5)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw)
    got, counts = gpu_frames(raw)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want)
    acc, _ = gpu_frames(raw, emit=EMIT_ACCEPTED)
    assert_same_frames(acc, [c for c in want if c["accepted"]])
    p16 = synth.default_params(4, n_samples=300000, seed=6)
    p16.encoding = ENC_SC16
    raw = synth.fill_host(p16.n_samples, ENC_SC16, p16.seed, synth.make_frames(p16)).tobytes()
    got, _ = gpu_frames(raw, ENC_SC16)
    assert_same_frames(got, ob.scan(raw, ENC_SC16))


# ---- the two 8-bit scan kernels (SORA_ADSB_SCAN_VARIANT: 1, the default,
# magnitudes in registers, 7808-position tiles; 0 magnitudes through LDS,
# 16128-position tiles) give the same list on the same input.
SCAN_TILE_POS = {0: 16128, 1: 7808}


@pytest.mark.parametrize("variant", [0, 1])
def test_scan_variants_tile_edges_and_ragged_ends(monkeypatch, variant):
    monkeypatch.setenv("SORA_ADSB_SCAN_VARIANT", str(variant))
    monkeypatch.setenv("SORA_ADSB_SCAN_MIN_TILES", "2")
    with AdsbContext(1024) as ctx:
        assert ctx.scan_kernel_name(ENC_UC8) == ["adsb_scan_kernel<false>",
                                                 "adsb_scan_direct_kernel<256>"][variant]
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    T = SCAN_TILE_POS[variant]
    # frames that start just before, on and just after tile edges and the edges
    # of a wave's 63 lane runs (2016 positions) inside a tile
    edges = [T - 241, T + 1, 2 * T - 17, 2 * T + 2016 - 240, 3 * T + 2016 - 1,
             3 * T + 2 * 2016, 4 * T - 239, 5 * T + 31, 5 * T + 32 + 2016, 6 * T - 64,
             7 * T - 1, 8 * T]
    table = synth.frame_table([(p, klm, 70, 0, 112) for p in edges])
    for n in (8 * T + 239, 8 * T + 240, 8 * T + 241, 9 * T - 1, 9 * T, 9 * T + 1,
              9 * T - 1 + 240, 9 * T + 500):
        raw = synth.fill_host(n, ENC_UC8, 99, table).tobytes()
        want = ob.scan(raw)
        got, counts = gpu_frames(raw)
        assert_same_frames(got, want)
        assert counts["candidates"] == len(want)
    assert set(edges) <= {c["position"] for c in got if c["accepted"]}


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("enc", [ENC_UC8, ENC_SC8])
def test_scan_variants_noise_dense_hits_and_correction(monkeypatch, variant, enc):
    monkeypatch.setenv("SORA_ADSB_SCAN_VARIANT", str(variant))
    rng = np.random.default_rng(1000 + variant)
    raw = rng.integers(0, 256, size=2 * 200003, dtype=np.uint8).tobytes()
    got, _ = gpu_frames(raw, enc)
    assert_same_frames(got, ob.scan(raw, enc))
    if enc != ENC_UC8:
        return
    # a preamble every 20 samples: hits in every lane run
    p = synth.default_params(2, n_samples=150000, mean_gap=20, reply_permille=500,
                             unknown_permille=300, n_icao=9, seed=12)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw)
    got, counts = gpu_frames(raw)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want)
    # single-bit correction (parity unpinned, see above) in each kernel's phase 4
    p = synth.default_params(4, n_samples=600000, seed=13)     # 20 % flipped-bit frames
    p.encoding = ENC_UC8
    p.amp_min, p.amp_max = 30, 110
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw, correct=True)
    n_fixed = sum(1 for c in want if c["corrected_bit"] >= 0)
    assert n_fixed > 30
    got, counts = gpu_frames(raw, correct_single_bit=True)
    assert_same_frames(got, want)
    assert counts["corrected"] == n_fixed


@pytest.mark.parametrize("variant", [0, 1])
def test_long_scan_regions(monkeypatch, variant):
    """One workgroup for the whole capture: the longest range of tiles a scan
    region can be.  A candidate record carries its position within its TILE (6
    bits in the key, 8 in the tail); which tile, the filter finds in the range's
    tile_end[] slice, which it keeps in LDS -- SORA_RANGE_TILES_MAX slots, which is
    what limits a range (the scan refuses a longer one, it does not wrap)."""
    monkeypatch.setenv("SORA_ADSB_SCAN_VARIANT", str(variant))
    monkeypatch.setenv("SORA_ADSB_SCAN_MIN_TILES", "1000000")
    p = synth.default_params(2, n_samples=3000000, mean_gap=3000, reply_permille=200,
                             unknown_permille=300, n_icao=7, seed=31)
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    want = ob.scan(raw)
    got, counts = gpu_frames(raw)
    assert_same_frames(got, want)
    assert counts["candidates"] == len(want)
    assert got[-1]["position"] > 2900000
    acc, _ = gpu_frames(raw, emit=EMIT_ACCEPTED)
    assert_same_frames(acc, [c for c in want if c["accepted"]])
    # 2^22 positions and more in one range
    n = (1 << 22) + 20000
    buf = DeviceBuffer.from_host(bytes(2 * n + 16))
    with AdsbContext(n) as ctx:
        with pytest.raises(sora_amd.SoraError) as e:
            ctx.submit_device(buf, 2 * n, ENC_UC8)
        assert "too long" in str(e.value)


def test_randomized_parity_slice():
    """A seeded slice of tools/fuzz_parity.py in the driver's tier: captures of
    random length, density, encoding, emit mode and scan geometry, lists that have
    to grow, a dense stretch somewhere -- each against the oracle, every field."""
    import os
    rng = np.random.default_rng(20261111)
    dense7 = np.array([9, 1, 8, 1, 2, 2, 2], dtype=np.int64)
    saved = {k: os.environ.get(k) for k in ("SORA_ADSB_SCAN_ROUNDS", "SORA_ADSB_SCAN_MIN_TILES")}
    try:
        for case in range(60):
            enc = [ENC_UC8, ENC_UC8, ENC_SC8, ENC_SC16][rng.integers(0, 4)]
            n = int(rng.choice([rng.integers(240, 3000), rng.integers(3000, 100000),
                                rng.integers(100000, 600000)]))
            gap = int(rng.choice([60, 300, 1500, 6000]))
            p = synth.default_params(4 if enc == ENC_SC16 else 2, n_samples=n, mean_gap=gap,
                                     reply_permille=int(rng.integers(0, 500)),
                                     unknown_permille=int(rng.integers(0, 500)),
                                     n_icao=int(rng.integers(1, 40)),
                                     seed=int(rng.integers(1, 1 << 30)))
            p.encoding = enc
            raw = synth.fill_host(p.n_samples, enc, p.seed, synth.make_frames(p)).tobytes()
            if enc != ENC_SC16 and rng.integers(0, 4) == 0:
                m = int(rng.integers(1000, 30000))
                iq = np.full(2 * m, 128, dtype=np.uint8)
                iq[0::2] = 128 + 10 * dense7[np.arange(m) % 7]
                if enc == ENC_SC8:
                    iq = (iq.astype(np.int16) - 128).astype(np.int8).view(np.uint8)
                at = int(rng.integers(0, n)) * 2
                raw = raw[:at] + iq.tobytes() + raw[at:]
                n = len(raw) // 2
            os.environ["SORA_ADSB_SCAN_ROUNDS"] = str(int(rng.choice([1, 2, 16, 32, 64])))
            os.environ["SORA_ADSB_SCAN_MIN_TILES"] = str(int(rng.choice([1, 2, 8, 50])))
            emit = [EMIT_ACCEPTED, EMIT_CANDIDATES][rng.integers(0, 2)]
            fix = bool(rng.integers(0, 4) == 0)
            small = rng.integers(0, 3) == 0
            kw = dict(max_candidates=64, max_squitters=64, max_frames=64) if small else \
                dict(max_candidates=n // 2 + 1024, max_squitters=n // 2 + 1024)
            want = ob.scan(raw, enc, correct=fix)
            if emit == EMIT_ACCEPTED:
                want = [c for c in want if c["accepted"]]
            with AdsbContext(max(n, 1), emit_mode=emit, correct_single_bit=fix, **kw) as ctx:
                ctx.submit_host(raw, enc)
                got = frames_as_dicts(ctx.collect())
            try:
                assert_same_frames(got, want)
            except AssertionError as e:
                raise AssertionError(f"case {case}: enc {enc} n {n} gap {gap} emit {emit} "
                                     f"fix {fix} small {small} seed {p.seed}: {e}") from None
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_regrow_cases_on_poisoned_lists():
    """The list-regrow tests once more on the -DSORA_POISON_LISTS build of the
    library (`make -C sora_amd/csrc poison`): every list comes out of hipMalloc
    filled with 0xa5, so a kernel that reads a list nothing has written since
    produces frames that differ instead of passing on what the allocator left."""
    import os
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(sora_amd.__file__), "lib", "libsora_adsb_gpu_poison.so")
    if not os.path.exists(lib) or os.environ.get("SORA_ADSB_LIB"):
        pytest.skip("no poisoned build (or already running on one)")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__,
                        "-k", "overflow_reruns or regrow_keeps or record_ring",
                        "--timeout", "300", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                       env=dict(os.environ, SORA_ADSB_LIB=lib), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_two_contexts_scan_stream_and_resolve_streams():
    """sora_adsb_set_resolve_stream: two contexts take turns on ONE scan stream,
    each with a resolve stream of its own (bench.py --schedule two-stream): a
    context's calls keep their order across its two streams -- scan, squitter
    export, add_known, resolve, collect, then the reset and the scan of its next
    capture -- while the other context's scan runs beside its resolve.  Different
    captures alternate so that a result taken from the wrong scan would show."""
    import torch
    caps = []
    for seed in (11, 12, 13):
        p = synth.default_params(2, n_samples=700000 + 1000 * seed, mean_gap=600,
                                 reply_permille=300, unknown_permille=200, n_icao=6, seed=seed)
        raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
        d = torch.from_numpy(np.frombuffer(raw + bytes(16), dtype=np.uint8).copy()).cuda()
        caps.append((d, len(raw), [c for c in ob.scan(raw) if c["accepted"]]))
    scan_stream = torch.cuda.Stream()
    rstreams = [torch.cuda.Stream(priority=-1) for _ in range(2)]
    ctxs = [AdsbContext(720000, stream=scan_stream.cuda_stream,
                        resolve_stream=rs.cuda_stream, max_squitters=100000)
            for rs in rstreams]
    torch.cuda.synchronize()
    pending = None
    for i in range(9):
        ctx = ctxs[i % 2]
        d, nbytes, want = caps[i % 3]
        ctx.reset()
        ctx.scan(d, nbytes, ENC_UC8)
        ctx.resolve()
        if pending is not None:
            pctx, pwant = pending
            assert_same_frames(frames_as_dicts(pctx.collect()), pwant)
        pending = (ctx, want)
    pctx, pwant = pending
    assert_same_frames(frames_as_dicts(pctx.collect()), pwant)
    # back on one stream: the same context, the same answer
    ctxs[0].set_resolve_stream(None)
    d, nbytes, want = caps[1]
    ctxs[0].reset()
    ctxs[0].submit_device(d, nbytes, ENC_UC8)
    assert_same_frames(frames_as_dicts(ctxs[0].collect()), want)
    # streaming ingest with a resolve stream set
    ctxs[1].set_resolve_stream(rstreams[1].cuda_stream)
    ctxs[1].reset()
    raw = caps[2][0][:caps[2][1]].cpu().numpy().tobytes()
    got = frames_as_dicts(np.concatenate(list(ctxs[1].stream(raw, ENC_UC8, 200000))))
    assert_same_frames(got, caps[2][2])
    for c in ctxs:
        c.close()
