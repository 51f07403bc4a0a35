"""CPU tier: the scan kernel's per-lane code (sora_amd/csrc/adsb_tile.h), run by
the lane emulator, against the oracle."""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import GOLDEN_NAMES, load_golden


def emul_records(emulator, raw, encoding=1, emit_all=1, n_scan=None):
    buf = np.frombuffer(raw, dtype=np.uint8)
    n = buf.size // 2
    if n_scan is None:
        n_scan = max(0, n - 239)
    rec = np.zeros(max(16, n), dtype=np.uint64)
    k = emulator.emul_scan8(buf.ctypes.data, n, n_scan, encoding, emit_all,
                            rec.ctypes.data, rec.size)
    assert k < (1 << 62), "emulator invariant violated"
    return rec[:k]


def emul_records16(emulator, raw, emit_all=1, n_scan=None):
    buf = np.frombuffer(raw, dtype=np.uint8)
    n = buf.size // 4
    if n_scan is None:
        n_scan = max(0, n - 239)
    rec = np.zeros(max(16, n), dtype=np.uint64)
    k = emulator.emul_scan16(buf.ctypes.data, n, n_scan, emit_all, rec.ctypes.data, rec.size)
    assert k < (1 << 62), "emulator invariant violated"
    return rec[:k]


def sc16_noise(seed, n, sigma):
    """int16 I/Q noise; sigma 8000 puts I^2+Q^2 beyond 2^24, where the float
    rounding of the sc16 magnitude definition creates ties between different
    integers (SURVEY.md section 8a, sc16 note)."""
    rng = np.random.default_rng(seed)
    v = np.clip(np.rint(rng.normal(0, sigma, size=2 * n)), -32768, 32767)
    return v.astype("<i2").tobytes()


def check_against_oracle(records, cands):
    assert len(records) == len(cands)
    for r, c in zip(records, cands):
        r = int(r)
        kind, pos, addr = r >> 60, (r >> 24) & 0xFFFFFFFFF, r & 0xFFFFFF
        squitter = c["df"] in (11, 17, 18)
        want_kind = 0 if not squitter else (1 if c["syndrome"] == 0 else 2)
        assert pos == c["position"] and kind == want_kind
        assert addr == (c["address"] if want_kind == 1 else c["syndrome"])


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_emulator_on_golden(emulator, name):
    raw, _, _ = load_golden(name)
    check_against_oracle(emul_records(emulator, raw), ob.scan(raw))


@pytest.mark.parametrize("enc", [ob.ENC_UC8, ob.ENC_SC8])
def test_emulator_multi_tile_noise(emulator, enc):
    rng = np.random.default_rng(11 + enc)
    raw = rng.integers(0, 256, size=2 * 70001, dtype=np.uint8).tobytes()   # 5 tiles, ragged
    check_against_oracle(emul_records(emulator, raw, enc), ob.scan(raw, enc))


def test_emulator_frames_across_tile_edges(emulator):
    from sora_amd import synth
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    T = 16128           # SORA_TILE_POS; one frame per tile edge, none overlapping
    edges = [T - 241, T + 10, 2 * T - 17, 3 * T - 1, 4 * T, 5 * T + 1, 6 * T - 64,
             6 * T + 256, 7 * T - 239]
    table = synth.frame_table([(p, klm, 70, 0, 112) for p in edges])
    raw = synth.fill_host(7 * T + 500, ob.ENC_UC8, 99, table).tobytes()
    cands = ob.scan(raw)
    hit = {c["position"] for c in cands if c["accepted"]}
    assert set(edges) <= hit
    check_against_oracle(emul_records(emulator, raw), cands)


def test_emulator_owned_prefix(emulator):
    """n_owned < n: only the owned positions are scanned (middle shard)."""
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, size=2 * 40000, dtype=np.uint8).tobytes()
    cands = [c for c in ob.scan(raw) if c["position"] < 20000]
    check_against_oracle(emul_records(emulator, raw, n_scan=20000), cands)


@pytest.mark.parametrize("sigma", [3, 400, 8000, 30000])
def test_emulator_sc16_noise(emulator, sigma):
    """adsb_scan_kernel<true>'s lane code: 5 ragged tiles of sc16 noise (parity
    unpinned by the reference: the oracle's sc16 definition is the build's own)."""
    raw = sc16_noise(100 + sigma, 36001, sigma)
    cands = ob.scan(raw, ob.ENC_SC16)
    assert len(cands) > 300
    check_against_oracle(emul_records16(emulator, raw), cands)


def test_emulator_sc16_frames_across_tile_edges(emulator):
    from sora_amd import synth
    klm = synth.hex_to_bits("8D4840D6202CC371C32CE0576098")
    T = 7936            # SORA16_TILE_POS
    edges = [T - 241, T + 10, 2 * T - 17, 3 * T - 1, 4 * T, 5 * T + 1, 6 * T - 64,
             6 * T + 256, 7 * T - 239]
    table = synth.frame_table([(p, klm, 9000, 0, 112) for p in edges])
    raw = synth.fill_host(7 * T + 500, ob.ENC_SC16, 99, table).tobytes()
    cands = ob.scan(raw, ob.ENC_SC16)
    hit = {c["position"] for c in cands if c["accepted"]}
    assert set(edges) <= hit
    check_against_oracle(emul_records16(emulator, raw), cands)
    cands = [c for c in cands if c["position"] < 3 * T + 100]
    check_against_oracle(emul_records16(emulator, raw, n_scan=3 * T + 100), cands)


def test_emulator_dense_hits(emulator):
    """A hit at every tenth position (see tests/test_gpu_parity.py): many hits
    per lane, in order."""
    v = np.array([9, 1, 8, 1, 2, 2, 2, 5, 1, 3], dtype=np.int64)
    n = 2 * 16128 + 500
    iq = np.full(2 * n, 128, dtype=np.uint8)
    iq[0::2] = 128 + 10 * v[np.arange(n) % 10]
    raw = iq.tobytes()
    cands = ob.scan(raw)
    assert len(cands) > 3000
    check_against_oracle(emul_records(emulator, raw), cands)
