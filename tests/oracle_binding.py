"""ctypes bindings for the CPU oracle (oracle/adsb_oracle.c) and the compiled
reference (oracle/_ref).  Test infrastructure: only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libadsb_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

ENC_UC8, ENC_SC16, ENC_SC8 = 1, 2, 3
BYTES_PER_SAMPLE = {ENC_UC8: 2, ENC_SC8: 2, ENC_SC16: 4}


class OracleFrame(C.Structure):
    _fields_ = [
        ("position", C.c_uint64),
        ("bits", C.c_char * 113),
        ("amplitude_56", C.c_double),
        ("amplitude", C.c_double),
        ("df", C.c_int32),
        ("long_message", C.c_int32),
        ("syndrome", C.c_uint32),
        ("address", C.c_uint32),
        ("accepted", C.c_int32),
        ("corrected_bit", C.c_int32),
    ]


_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.oracle_magnitude.restype = C.c_size_t
        L.oracle_magnitude.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.oracle_preamble.restype = C.c_int
        L.oracle_preamble.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_crc24.restype = C.c_uint32
        L.oracle_crc24.argtypes = [C.c_char_p, C.c_int]
        L.oracle_plane_set_new.restype = C.c_void_p
        L.oracle_plane_set_delete.argtypes = [C.c_void_p]
        L.oracle_plane_set_count.restype = C.c_size_t
        L.oracle_plane_set_count.argtypes = [C.c_void_p]
        L.oracle_scan.restype = C.c_size_t
        L.oracle_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p,
                                  C.c_int, C.c_void_p, C.c_size_t]
        L.oracle_count_candidates.restype = C.c_size_t
        L.oracle_count_candidates.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_parse_message.restype = C.c_int
        L.oracle_parse_message.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong,
                                           C.c_longlong, C.c_double, C.c_void_p]
        L.oracle_print_last.restype = C.c_int
        L.oracle_print_last.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.oracle_set_correction.argtypes = [C.c_int]
        L.oracle_correctable_bit.restype = C.c_int
        L.oracle_correctable_bit.argtypes = [C.c_uint32, C.c_int]
        _lib = L
    return _lib


def magnitude(iq_bytes, encoding=ENC_UC8):
    """float32 |z|^2 per sample (row a1)."""
    buf = np.ascontiguousarray(np.frombuffer(iq_bytes, dtype=np.uint8))
    n = buf.size // BYTES_PER_SAMPLE[encoding]
    m = np.empty(max(n, 1), dtype=np.float32)
    got = lib().oracle_magnitude(buf.ctypes.data, buf.size, encoding, m.ctypes.data)
    assert got == n
    return m[:n]


class PlaneSet:
    def __init__(self):
        self.h = lib().oracle_plane_set_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_plane_set_delete(self.h)
            self.h = None

    def __len__(self):
        return lib().oracle_plane_set_count(self.h)

    def parse(self, bits, sec=0, usec=0, amplitude=0.0):
        f = OracleFrame()
        ok = lib().oracle_parse_message(self.h, bits.encode(), sec, usec, amplitude,
                                        C.byref(f))
        return bool(ok), f

    def print_last(self):
        buf = C.create_string_buffer(512)
        n = lib().oracle_print_last(self.h, buf, 512)
        return buf.raw[:n].decode()


def scan(iq_bytes, encoding=ENC_UC8, accepted_only=False, base_position=0,
         plane_set=None, cap=None, correct=False):
    """The whole chain on a capture: list of frame dicts in stream order.
    correct=True turns on the single-bit correction extension (D3)."""
    m = magnitude(iq_bytes, encoding)
    ps = plane_set or PlaneSet()
    if cap is None:
        cap = max(1024, m.size // 2 + 16)
    arr = (OracleFrame * cap)()
    lib().oracle_set_correction(1 if correct else 0)
    try:
        n = lib().oracle_scan(m.ctypes.data, m.size, base_position, ps.h,
                              1 if accepted_only else 0, arr, cap)
    finally:
        lib().oracle_set_correction(0)
    assert n <= cap, "oracle frame buffer too small"
    out = []
    for i in range(n):
        f = arr[i]
        out.append(dict(position=f.position, bits=f.bits.decode(),
                        amplitude_56=f.amplitude_56, amplitude=f.amplitude,
                        df=f.df, long=f.long_message, syndrome=f.syndrome,
                        address=f.address, accepted=bool(f.accepted),
                        corrected_bit=f.corrected_bit))
    return out


def main_loop_text(iq_bytes, flags, encoding=ENC_UC8, t0=1700000000, rate=2000000,
                   correct=False):
    """planes_main_loop output of the oracle, as text (TZ from environment)."""
    import tempfile
    L = lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.oracle_main_loop.restype = C.c_size_t
    L.oracle_main_loop.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_longlong,
                                   C.c_ulonglong, C.c_void_p]
    m = magnitude(iq_bytes, encoding)
    with tempfile.NamedTemporaryFile(suffix=".txt", delete=False) as tf:
        path = tf.name
    try:
        fp = libc.fopen(path.encode(), b"w")
        L.oracle_set_correction(1 if correct else 0)
        L.oracle_main_loop(m.ctypes.data, m.size, flags, t0, rate, fp)
        L.oracle_set_correction(0)
        libc.fclose(fp)
        with open(path) as fh:
            return fh.read()
    finally:
        os.unlink(path)


# ---------------------------------------------------------------- reference --

def have_reference():
    return os.path.exists(os.path.join(REF_DIR, "ref_candidates"))


def ref_candidates(iq_bytes, optimized=True, accepted_only=False):
    """Every candidate of the COMPILED REFERENCE with its position."""
    exe = os.path.join(REF_DIR, "ref_candidates_O2" if optimized else "ref_candidates")
    args = [exe] + (["--accepted-only"] if accepted_only else [])
    p = subprocess.run(args, input=bytes(iq_bytes), capture_output=True, check=True)
    out = []
    for line in p.stdout.decode().splitlines():
        pos, bits, a56, amp, acc = line.split()
        out.append(dict(position=int(pos), bits=bits,
                        amplitude_56=float.fromhex(a56),
                        amplitude=float.fromhex(amp), accepted=acc == "1"))
    final = None
    for line in p.stderr.decode().splitlines():
        if line.startswith("final_position"):
            final = int(line.split()[1])
    return out, final


def ref_main_loop(iq_bytes, bitstring=False, shim=True, env=None, from_raw=True):
    """stdout of the compiled reference chain (`sora --adsb-decode --adsb-from-raw`;
    from_raw=False: the ASCII bit-string input mode)."""
    exe = os.path.join(REF_DIR, "ref_sora_adsb_shim" if shim else "ref_sora_adsb")
    args = [exe, "--adsb-decode"] + (["--adsb-from-raw"] if from_raw else [])
    if bitstring:
        args.append("--adsb-to-bitstring")
    e = dict(os.environ)
    e.setdefault("TZ", "UTC")
    if env:
        e.update(env)
    p = subprocess.run(args, input=bytes(iq_bytes), capture_output=True, check=True,
                       env=e)
    return p.stdout.decode()
