"""CPU tier: the oracle against the reference's behaviour.

1. against tests/golden/ (vectors the COMPILED REFERENCE produced, see
   tools/make_golden.py) -- runs everywhere;
2. against the compiled reference itself on fresh random input -- runs where
   oracle/_ref exists (the build container).
"""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import GOLDEN_NAMES, assert_same_frames, load_golden


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_matches_golden_candidates(name):
    raw, cands, _ = load_golden(name)
    got = ob.scan(raw)
    assert_same_frames(got, cands, fields=("position", "bits", "amplitude_56",
                                            "amplitude", "accepted"))


@pytest.mark.parametrize("name", GOLDEN_NAMES)
@pytest.mark.parametrize("mode,flags", [("decoded", 0x10), ("bitstring", 0x20)])
def test_oracle_matches_golden_text(name, mode, flags):
    raw, _, text = load_golden(name)
    assert ob.main_loop_text(raw, flags) == text[mode]


def test_survey_facts():
    """SURVEY.md section 4 facts, as reproduced by the compiled reference."""
    raw, cands, text = load_golden("klm_n1240")
    assert len(cands) == 1 and cands[0]["position"] == 1000 and cands[0]["accepted"]
    assert cands[0]["bits"] == format(int("8D4840D6202CC371C32CE0576098", 16), "0112b")
    assert cands[0]["amplitude"] == 0.4542236328125
    assert "4840D6 KLM1023" in text["decoded"]
    assert text["bitstring"].rstrip().endswith(",0.454224")
    raw, cands, _ = load_golden("klm_n1239")        # one sample short: dropped
    assert cands == []
    raw, cands, _ = load_golden("df11_df17")
    assert [c["position"] for c in cands] == [500, 1500, 2500]
    assert all(len(c["bits"]) == 112 for c in cands)    # short frames still 112 (D2)


def test_predicate_rate_on_uniform_noise():
    rng = np.random.default_rng(7)
    iq = rng.integers(0, 256, size=2 * 400000, dtype=np.uint8).tobytes()
    m = ob.magnitude(iq)
    hits = ob.lib().oracle_count_candidates(m.ctypes.data, m.size)
    assert abs(hits / (m.size - 239) - 0.0252) < 0.001      # 2.52 %


def test_short_inputs():
    for n in (0, 1, 17, 18, 239):
        assert ob.scan(bytes(2 * n)) == []


def test_crc_known_answers():
    L = ob.lib()
    klm = format(int("8D4840D6202CC371C32CE0576098", 16), "0112b")
    assert L.oracle_crc24(klm.encode(), 88) == int(klm[88:], 2)
    df11 = format(int("5D4840D6F8740F", 16), "056b")
    assert L.oracle_crc24(df11.encode(), 32) == int(df11[32:], 2)


def test_stateful_rule_needs_an_earlier_squitter():
    """plane.c:479-483: a reply is accepted only once its address is known."""
    from sora_amd import synth, bits_to_string
    icao = 0x4840D6
    reply = bits_to_string(synth.build_short_reply(5, 0x1234567, icao))
    squit = bits_to_string(synth.build_df11(icao))
    ps = ob.PlaneSet()
    assert ps.parse(reply)[0] is False
    assert ps.parse(squit)[0] is True
    ok, info = ps.parse(reply)
    assert ok and info.address == icao and len(ps) == 1


@pytest.mark.skipif(not ob.have_reference(), reason="compiled reference not present")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_matches_reference_live(seed):
    from sora_amd import synth
    p = synth.default_params(2, n_samples=300000, mean_gap=900, reply_permille=200,
                             unknown_permille=300, n_icao=5, seed=seed,
                             first_position=100)
    iq = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    ref, final = ob.ref_candidates(iq)
    assert final == p.n_samples - 17
    got = ob.scan(iq)
    assert_same_frames(got, ref, fields=("position", "bits", "amplitude_56", "amplitude",
                                          "accepted"))
    assert sum(c["accepted"] for c in ref) > 20
    env = {"TZ": "UTC", "SORA_T0": "1700000000", "SORA_RATE": "2000000"}
    assert ob.main_loop_text(iq, 0x10) == ob.ref_main_loop(iq, False, True, env)
    assert ob.main_loop_text(iq, 0x20) == ob.ref_main_loop(iq, True, True, env)


def test_correction_extension_is_off_by_default_and_round_trips():
    """Extension D3, parity unpinned: every single
    flipped bit b >= 5 of a clean DF17 / DF11 is found again; DF bits are not."""
    from sora_amd import synth
    L = ob.lib()
    for hexmsg, nbits in (("8D4840D6202CC371C32CE0576098", 112), ("5D4840D6F8740F", 56)):
        clean = "".join(format(w, "032b") for w in synth.hex_to_bits(hexmsg))[:112]
        ps = ob.PlaneSet()
        ok, f = ps.parse(clean)
        assert ok and f.syndrome == 0
        for b in range(nbits):
            flipped = clean[:b] + ("1" if clean[b] == "0" else "0") + clean[b + 1:]
            ok, f = ob.PlaneSet().parse(flipped)
            if b < 5:
                continue                      # another DF: not a squitter any more
            assert not ok
            assert L.oracle_correctable_bit(f.syndrome, nbits) == b
    assert L.oracle_correctable_bit(0, 112) == -1
    assert L.oracle_correctable_bit(0x123456, 112) == -1
    # x^107 mod G is the syndrome of bit 4 of a long message: inside the DF field
    v = 1
    for _ in range(107):
        v <<= 1
        if v & 0x1000000:
            v ^= 0x1FFF409
    assert L.oracle_correctable_bit(v, 112) == -1
    # scan(): default output is unchanged by the existence of the extension
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 256, size=2 * 50000, dtype=np.uint8).tobytes()
    assert all(c["corrected_bit"] == -1 for c in ob.scan(raw))
