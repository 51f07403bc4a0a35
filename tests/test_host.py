"""The reference's C entry points (include/sora_adsb_host.h) as built in
libsora_adsb_host.so / bin/sora.  CPU tier: symbols, the validator + printer
through the ASCII input mode (no device involved) against what the compiled
reference printed.  GPU tier: `sora --adsb-decode --adsb-from-raw` end to end."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, GOLDEN_NAMES, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SORA = os.path.join(ROOT, "bin", "sora")
HOST_SO = os.path.join(ROOT, "sora_amd", "lib", "libsora_adsb_host.so")
ENV = dict(os.environ, TZ="UTC", SORA_T0="1700000000", SORA_RATE="2000000")


def run_sora(args, data, env=None):
    e = dict(ENV)
    if env:
        e.update(env)
    p = subprocess.run([SORA] + args, input=data, capture_output=True, env=e, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def test_host_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "sora_adsb_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b((?:plane|planes|radio)_[a-z0-9_]+)\s*\(", text))
    assert len(names) > 30
    # the reference symbols the path is entered through
    assert {"plane_iq_state_new", "plane_iq_get_next", "plane_iq_state_delete",
            "planes_main_loop", "plane_set_parse_message", "radio_file_open"} <= names
    L = C.CDLL(HOST_SO)
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in sora_adsb_host.h but not exported"


@pytest.mark.parametrize("name", [n for n in GOLDEN_NAMES if n != "uniform_noise_odd"])
def test_ascii_mode_matches_reference(name):
    """planes-main-loop.c:15-47 + plane.c:457-632: validator, decoder, printer."""
    with open(os.path.join(GOLDEN, name + ".bitstring.txt"), "rb") as fh:
        data = fh.read()
    with open(os.path.join(GOLDEN, name + ".decoded_from_ascii.txt")) as fh:
        want = fh.read()
    assert run_sora(["--adsb-decode"], data) == want
    # bit-string output of bit-string input: amplitude goes through %g once more
    again = run_sora(["--adsb-decode", "--adsb-to-bitstring"], data)
    assert [l.split(" @ ")[0] for l in again.splitlines()] == \
           [l.split(" @ ")[0] for l in data.decode().splitlines()]


def test_validator_rejects_what_the_reference_rejects():
    """An unknown address is dropped until a clean squitter introduces it."""
    from sora_amd import bits_to_string, synth
    icao = 0xABCDEF
    reply = bits_to_string(synth.build_short_reply(4, 0x0000310, icao))
    squit = bits_to_string(synth.build_df11(icao))
    bad = squit[:40] + ("1" if squit[40] == "0" else "0") + squit[41:]
    lines = "".join(f"{m} @ 1700000000,0,0,0.5\n" for m in (reply, bad, squit, reply))
    out = run_sora(["--adsb-decode"], lines.encode()).splitlines()
    assert out[0].startswith("---") and len(out) == 3        # header, squitter, reply
    assert out[1].startswith("ABCDEF") and out[2].startswith("ABCDEF")


def test_cli_errors():
    p = subprocess.run([SORA, "--scan"], capture_output=True)
    assert p.returncode != 0
    p = subprocess.run([SORA, "-d", "--file=/nonexistent/x.uc8"], capture_output=True)
    assert p.returncode != 0 and b"can't open file" in p.stderr


def test_no_gpu_fails_loudly():
    import sora_amd
    if sora_amd._lib.lib().sora_adsb_device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([SORA, "-d", "--adsb-from-raw"], input=bytes(4096),
                       capture_output=True, env=ENV)
    assert p.stdout == b"" and b"cannot use the GPU ADS-B path" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_NAMES)
@pytest.mark.parametrize("block", [None, "20000"])
def test_sora_from_raw_matches_reference(name, block):
    raw, _, text = load_golden(name)
    env = {"SORA_ADSB_BLOCK_SAMPLES": block} if block else None
    assert run_sora(["--adsb-decode", "--adsb-from-raw"], raw, env) == text["decoded"]
    assert run_sora(["-d", "--adsb-from-raw", "--adsb-to-bitstring"], raw, env) == \
        text["bitstring"]


@pytest.mark.gpu
def test_sora_file_option_and_emit_all(tmp_path):
    raw, _, text = load_golden("mixed_traffic")
    path = tmp_path / "capture.uc8"
    path.write_bytes(raw)
    out = run_sora(["-d", "--adsb-from-raw", f"--file={path}", "--file-encoding=uc8"], b"")
    assert out == text["decoded"]
    # every candidate through the iterator, the validator decides on the host
    out = run_sora(["-d", "--adsb-from-raw"], raw, {"SORA_ADSB_EMIT_ALL": "1"})
    assert out == text["decoded"]
    out = run_sora(["-d", "--adsb-from-raw", "--file=-"], raw)
    assert out == text["decoded"]


@pytest.mark.gpu
@pytest.mark.parametrize("encname", ["sc8", "sc16"])
def test_sora_file_encodings_match_oracle(tmp_path, encname):
    """--file / --file-encoding routing for sc8 and sc16.
    Parity unpinned by the reference (it has no such route): the oracle's main
    loop on the same samples is the expected text."""
    import oracle_binding as ob
    from sora_amd import synth
    enc = {"sc8": ob.ENC_SC8, "sc16": ob.ENC_SC16}[encname]
    p = synth.default_params(4 if encname == "sc16" else 2, n_samples=400000, mean_gap=800,
                             reply_permille=250, unknown_permille=200, n_icao=5, seed=12)
    p.encoding = enc
    raw = synth.fill_host(p.n_samples, enc, p.seed, synth.make_frames(p)).tobytes()
    path = tmp_path / f"capture.{encname}"
    path.write_bytes(raw)
    for flags, extra in ((0x11, []), (0x21, ["--adsb-to-bitstring"])):
        want = ob.main_loop_text(raw, flags, enc)
        assert want.count("\n") > 50
        got = run_sora(["-d", "--adsb-from-raw", f"--file={path}",
                        f"--file-encoding={encname}"] + extra, b"",
                       {"SORA_ADSB_BLOCK_SAMPLES": "70000"})
        assert got == want


@pytest.mark.gpu
def test_sora_correction_env_matches_oracle():
    """SORA_ADSB_CORRECT=1 (section 8 f.3): corrected squitters reach the plane set
    and the printer exactly as in the oracle's main loop with correction on."""
    import oracle_binding as ob
    from sora_amd import synth
    p = synth.default_params(4, n_samples=600000, seed=4)       # 20 % one-bit errors
    p.encoding = ob.ENC_UC8
    p.amp_min, p.amp_max = 30, 110
    raw = synth.fill_host(p.n_samples, p.encoding, p.seed, synth.make_frames(p)).tobytes()
    plain = ob.main_loop_text(raw, 0x21)
    fixed = ob.main_loop_text(raw, 0x21, correct=True)
    assert fixed.count("\n") > plain.count("\n") + 50
    args = ["-d", "--adsb-from-raw", "--adsb-to-bitstring"]
    assert run_sora(args, raw) == plain
    assert run_sora(args, raw, {"SORA_ADSB_CORRECT": "1"}) == fixed
    assert run_sora(["-d", "--adsb-from-raw"], raw, {"SORA_ADSB_CORRECT": "1"}) == \
        ob.main_loop_text(raw, 0x11, correct=True)


def test_radio_file_read_conversions(tmp_path):
    """Row a14 (radio-file.c:96-152): every encoding widens to double complex
    exactly ((b-128)/128, b/128, s/32768; u8 and s16 real), whole items only, 0
    at the end of the file.  Expected values computed here in float64 (the
    arithmetic is exact in double)."""
    import numpy as np
    L = C.CDLL(HOST_SO)

    class Methods(C.Structure):
        pass

    class Radio(C.Structure):
        _fields_ = [("m", C.POINTER(Methods))]

    RP = C.POINTER(Radio)
    Methods._fields_ = [
        ("set_frequency", C.CFUNCTYPE(C.c_int, RP, C.c_uint64)),
        ("get_frequency", C.CFUNCTYPE(C.c_int, RP, C.POINTER(C.c_uint64))),
        ("set_sample_rate", C.CFUNCTYPE(C.c_int, RP, C.c_ulong)),
        ("get_sample_rate", C.CFUNCTYPE(C.c_int, RP, C.POINTER(C.c_ulong))),
        ("read", C.CFUNCTYPE(C.c_ssize_t, RP, C.c_void_p, C.c_size_t)),
        ("get_file_position", C.CFUNCTYPE(C.c_long, RP)),
        ("close", C.CFUNCTYPE(None, RP)),
    ]
    L.radio_file_open.restype = RP
    L.radio_file_open.argtypes = [C.c_char_p, C.c_int]
    rng = np.random.default_rng(7)
    raw = rng.integers(0, 256, size=4 * 1000 + 3, dtype=np.uint8)    # ragged tail
    path = tmp_path / "x.bin"
    path.write_bytes(raw.tobytes())
    u8 = raw.astype(np.float64)
    s8 = raw.view(np.int8).astype(np.float64)
    s16 = raw[:len(raw) // 2 * 2].view("<i2").astype(np.float64)
    want = {
        1: ((u8[0::2][:2001] - 128) / 128) + 1j * ((u8[1::2][:2001] - 128) / 128),   # uc8
        3: (s8[0::2][:2001] / 128) + 1j * (s8[1::2][:2001] / 128),                   # sc8
        2: (s16[0::2][:1000] / 32768) + 1j * (s16[1::2][:1000] / 32768),             # sc16
        32: ((u8 - 128) / 128) + 0j,                                                 # u8
        33: (s16 / 32768) + 0j,                                                      # s16
    }
    # the same file is a golden fixture: what the COMPILED REFERENCE's
    # radio->m->read() returned for it (oracle/ref_radio_file_driver.c,
    # tools/make_golden.py) must equal the float64 formulas above
    with open(os.path.join(GOLDEN, "radio_file.bin"), "rb") as fh:
        assert fh.read() == raw.tobytes()
    for enc, expect in want.items():
        with open(os.path.join(GOLDEN, f"radio_file.enc{enc}.txt")) as fh:
            ref = np.array([[float.fromhex(x) for x in line.split()] for line in fh])
        assert np.array_equal(ref[:, 0], expect.real) and np.array_equal(ref[:, 1], expect.imag)
    for enc, expect in want.items():
        r = L.radio_file_open(str(path).encode(), enc)
        assert r
        m = r.contents.m.contents
        got = []
        buf = np.empty(300, dtype=np.complex128)                 # several partial reads
        while True:
            k = m.read(r, buf.ctypes.data, buf.size)
            if k <= 0:
                break
            got.append(buf[:k].copy())
        rate = C.c_ulong()
        assert m.get_sample_rate(r, C.byref(rate)) == 0 and rate.value == 1   # radio-file.c:51
        m.close(r)
        got = np.concatenate(got)
        assert got.shape == expect.shape, enc
        assert np.array_equal(got.view(np.float64), expect.astype(np.complex128).view(np.float64))
    # an unknown encoding opens (main.c:229-230 only warns) and reads nothing
    r = L.radio_file_open(str(path).encode(), 0)
    if r:
        assert r.contents.m.contents.read(r, buf.ctypes.data, 10) == 0
        r.contents.m.contents.close(r)
