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
    """No device: nothing is decoded on the CPU, and the exit status says so."""
    import sora_amd
    if sora_amd._lib.lib().sora_adsb_device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([SORA, "-d", "--adsb-from-raw"], input=bytes(4096),
                       capture_output=True, env=ENV)
    assert p.stdout == b"" and b"cannot use the GPU ADS-B path" in p.stderr
    assert p.returncode != 0


REF_LOOP = os.path.join(ROOT, "oracle", "_ref", "ref_loop_on_gpu")


def run_ref_loop(args, data, env=None):
    """The reference's UNMODIFIED planes-main-loop.c (+ oracle/ref_main_driver.c)
    linked against libsora_adsb_host.so: oracle/Makefile, _ref/ref_loop_on_gpu."""
    if not os.path.exists(REF_LOOP):
        pytest.skip("oracle/_ref/ref_loop_on_gpu is not built (needs /root/reference)")
    e = dict(ENV)
    if env:
        e.update(env)
    p = subprocess.run([REF_LOOP] + args, input=data, capture_output=True, env=e, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


@pytest.mark.parametrize("name", [n for n in GOLDEN_NAMES if n != "uniform_noise_odd"])
def test_reference_loop_on_host_library_ascii(name):
    """INTEGRATION.md section 1 for the validator and printer (B3): the reference's
    own main loop calling this library's plane_set_* / plane_print gives the text
    the all-reference build printed."""
    with open(os.path.join(GOLDEN, name + ".bitstring.txt"), "rb") as fh:
        data = fh.read()
    with open(os.path.join(GOLDEN, name + ".decoded_from_ascii.txt")) as fh:
        want = fh.read()
    assert run_ref_loop(["--adsb-decode"], data) == want


def test_plane_set_iterate_and_radio_init():
    """plane.h:18 (util/iterator.h:4-14) and radio.h:15 (radio.c:18-86)."""
    L = C.CDLL(HOST_SO)

    class Iterator(C.Structure):
        pass

    IP = C.POINTER(Iterator)
    Iterator._fields_ = [
        ("is_at_end", C.CFUNCTYPE(C.c_int, IP)),
        ("next", C.CFUNCTYPE(C.c_void_p, IP)),
        ("next_pair", C.CFUNCTYPE(None, IP, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))),
        ("free", C.CFUNCTYPE(None, IP)),
    ]
    L.plane_set_new.restype = C.c_void_p
    L.plane_set_lookup_create.restype = C.c_void_p
    L.plane_set_lookup_create.argtypes = [C.c_void_p, C.c_uint32]
    L.plane_set_iterate.restype = IP
    L.plane_set_iterate.argtypes = [C.c_void_p]
    L.plane_set_delete.argtypes = [C.c_void_p]
    ps = L.plane_set_new()
    it = L.plane_set_iterate(ps)                        # empty set
    assert it.contents.is_at_end(it) == 1 and it.contents.next(it) is None
    it.contents.free(it)
    addresses = [0x4840D6, 0xABCDEF, 1, 0xFFFFFF] + list(range(1000, 1600))   # grows the table
    planes = {a: L.plane_set_lookup_create(ps, a) for a in addresses}
    assert L.plane_set_lookup_create(ps, 0xABCDEF) == planes[0xABCDEF]
    it = L.plane_set_iterate(ps)
    seen = []
    while not it.contents.is_at_end(it):
        seen.append(it.contents.next(it))
    it.contents.free(it)
    assert sorted(seen) == sorted(planes.values())
    it = L.plane_set_iterate(ps)
    pairs = {}
    while not it.contents.is_at_end(it):
        k, d = C.c_void_p(), C.c_void_p()
        it.contents.next_pair(it, C.byref(k), C.byref(d))
        pairs[k.value or 0] = d.value
    k, d = C.c_void_p(1), C.c_void_p(1)
    it.contents.next_pair(it, C.byref(k), C.byref(d))   # past the end: hash.c:385-386
    assert k.value is None and d.value is None
    it.contents.free(it)
    assert pairs == planes
    L.plane_set_delete(ps)

    # radio_init fills the empty slots of a vtable with the dummies of radio.c:46-86
    class Methods(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("set_frequency", "get_frequency",
                                              "set_sample_rate", "get_sample_rate", "read",
                                              "get_file_position", "close")]

    class Radio(C.Structure):
        _fields_ = [("m", C.POINTER(Methods))]

    m, r = Methods(), Radio()
    L.radio_init(C.byref(r), C.byref(m))
    assert all(getattr(m, n) for n, _ in Methods._fields_)
    assert C.addressof(r.m.contents) == C.addressof(m)
    rp = C.byref(r)
    assert C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)(m.set_frequency)(rp, 5) == -1
    assert C.CFUNCTYPE(C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_size_t)(m.read)(rp, None, 0) == -1
    assert C.CFUNCTYPE(C.c_long, C.c_void_p)(m.get_file_position)(rp) == 0
    C.CFUNCTYPE(None, C.c_void_p)(m.close)(rp)


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
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_reference_loop_on_host_library_from_raw(name):
    """INTEGRATION.md section 1 as a program: the reference's unmodified
    planes-main-loop.c on top of plane_iq_state_new / plane_iq_get_next (every
    candidate, the reference's iterator semantics) / plane_set_parse_message /
    plane_print of this library prints what the all-reference build printed."""
    raw, cands, text = load_golden(name)
    assert run_ref_loop(["--adsb-decode", "--adsb-from-raw"], raw) == text["decoded"]
    assert run_ref_loop(["-d", "--adsb-from-raw", "--adsb-to-bitstring"], raw,
                        {"SORA_ADSB_BLOCK_SAMPLES": "20000"}) == text["bitstring"]


@pytest.mark.gpu
def test_iterator_yields_every_candidate_by_default():
    """plane-iq.c:187-192: the iterator returns every completed candidate; the
    accepted-only form is this library's own main loop's."""
    from sora_amd import ENC_UC8
    raw, cands, _ = load_golden("mixed_traffic")
    L = C.CDLL(HOST_SO)
    libc = C.CDLL(None)
    libc.fmemopen.restype = C.c_void_p
    libc.fmemopen.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    class Timeval(C.Structure):
        _fields_ = [("tv_sec", C.c_long), ("tv_usec", C.c_long)]

    for fn in ("plane_iq_state_new", "plane_iq_state_new_encoded", "plane_iq_state_new_accepted"):
        getattr(L, fn).restype = C.c_void_p
    L.plane_iq_state_new.argtypes = [C.c_void_p]
    L.plane_iq_state_new_encoded.argtypes = [C.c_void_p, C.c_int]
    L.plane_iq_state_new_accepted.argtypes = [C.c_void_p, C.c_int]
    L.plane_iq_get_next.restype = C.c_char_p
    L.plane_iq_get_next.argtypes = [C.c_void_p, C.POINTER(Timeval), C.POINTER(C.c_double)]
    L.plane_iq_last_position.restype = C.c_ulonglong
    L.plane_iq_last_position.argtypes = [C.c_void_p]
    L.plane_iq_state_delete.argtypes = [C.c_void_p]
    L.plane_iq_error.argtypes = [C.c_void_p]
    os.environ.pop("SORA_ADSB_EMIT_ALL", None)
    buf = C.create_string_buffer(raw, len(raw))

    def pull(make):
        f = libc.fmemopen(buf, len(raw), b"rb")
        st = make(f)
        assert st
        out, tv, amp = [], Timeval(), C.c_double()
        while True:
            msg = L.plane_iq_get_next(st, C.byref(tv), C.byref(amp))
            if msg is None:
                break
            out.append((L.plane_iq_last_position(st), msg.decode(), amp.value))
        assert L.plane_iq_error(st) == 0
        L.plane_iq_state_delete(st)
        libc.fclose(f)
        return out

    want_all = [(c["position"], c["bits"], c["amplitude"]) for c in cands]
    want_acc = [(c["position"], c["bits"], c["amplitude"]) for c in cands if c["accepted"]]
    assert len(want_all) > 10 * len(want_acc) > 0
    assert pull(lambda f: L.plane_iq_state_new(f)) == want_all
    assert pull(lambda f: L.plane_iq_state_new_encoded(f, ENC_UC8)) == want_all
    assert pull(lambda f: L.plane_iq_state_new_accepted(f, ENC_UC8)) == want_acc


def dense_then_mixed():
    """A stretch with a preamble hit at every tenth position (test_gpu_parity's
    dense_candidates: 10 % of the positions are candidates, four times what
    uniform noise gives) followed by ordinary traffic."""
    import numpy as np
    n = 300000
    v = np.array([9, 1, 8, 1, 2, 2, 2, 5, 1, 3], dtype=np.int64)
    iq = np.full(2 * n, 128, dtype=np.uint8)
    iq[0::2] = 128 + 10 * v[np.arange(n) % 10]
    mixed, _, _ = load_golden("mixed_traffic")
    return iq.tobytes() + mixed


@pytest.mark.gpu
def test_sora_dense_candidates_do_not_end_the_stream():
    """The reference never drops a candidate (plane-iq.c:141-194).  Default
    settings, and then lists far too small for the block at hand: the library
    grows them and runs the block again, the text is the oracle's."""
    import oracle_binding as ob
    raw = dense_then_mixed()
    want = {flags: ob.main_loop_text(raw, flags) for flags in (0x11, 0x21)}
    assert want[0x11].count("\n") > 50
    for env in (None, {"SORA_ADSB_BLOCK_SAMPLES": "100000"},
                {"SORA_ADSB_BLOCK_SAMPLES": "100000", "SORA_ADSB_MAX_CANDIDATES": "64"},
                {"SORA_ADSB_BLOCK_SAMPLES": "70000", "SORA_ADSB_MAX_CANDIDATES": "64",
                 "SORA_ADSB_EMIT_ALL": "1"}):
        assert run_sora(["-d", "--adsb-from-raw"], raw, env) == want[0x11]
        assert run_sora(["-d", "--adsb-from-raw", "--adsb-to-bitstring"], raw, env) == want[0x21]
    assert run_ref_loop(["-d", "--adsb-from-raw"], raw,
                        {"SORA_ADSB_BLOCK_SAMPLES": "100000",
                         "SORA_ADSB_MAX_CANDIDATES": "64"}) == want[0x11]


@pytest.mark.gpu
def test_sora_device_failure_is_not_end_of_input():
    """A device that cannot be used must not look like an empty capture."""
    raw, _, _ = load_golden("df11_df17")
    p = subprocess.run([SORA, "-d", "--adsb-from-raw"], input=raw, capture_output=True,
                       env=dict(ENV, SORA_ADSB_DEVICE="99"))
    assert p.returncode != 0 and p.stdout == b""
    assert b"cannot use the GPU ADS-B path" in p.stderr and b"decoding failed" in p.stderr


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


@pytest.mark.parametrize("tier", ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_radio_file_read_conversions(tmp_path, tier):
    """(Host-only code; collected in both tiers so that the driver's -m gpu run on
    the GPU box exercises row a14 too.)  Row a14 (radio-file.c:96-152): every encoding widens to double complex
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
