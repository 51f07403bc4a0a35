"""ctypes binding of libsora_adsb_gpu.so (include/sora_adsb_gpu.h, sora_synth.h).

The library is the product: there is no Python or CPU fallback.  Importing this
module without the built library raises; calling it on a machine without a
gfx950 device returns the C error codes as SoraError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SORA_ADSB_LIB: another build of the same library (the test suite runs its
# list-regrow cases on the -DSORA_POISON_LISTS build this way)
LIB_PATH = os.environ.get("SORA_ADSB_LIB") or os.path.join(_HERE, "lib", "libsora_adsb_gpu.so")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OVERFLOW = -1, -2, -3, -4
ERR_NOMEM, ERR_ALIGN, ERR_STATE = -5, -6, -7

ENC_UC8, ENC_SC16, ENC_SC8 = 1, 2, 3
BYTES_PER_SAMPLE = {ENC_UC8: 2, ENC_SC8: 2, ENC_SC16: 4}
ENCODING_NAMES = {"uc8": ENC_UC8, "sc8": ENC_SC8, "sc16": ENC_SC16}

EMIT_ACCEPTED, EMIT_CANDIDATES = 0, 1
FRAME_LONG, FRAME_SQUITTER, FRAME_ACCEPTED, FRAME_CORRECTED = 1, 2, 4, 8
OPT_CORRECT_SINGLE_BIT = 1
OPT_FIXED_LISTS = 2
OPT_WIRE_FRAMES = 4
WINDOW, HALO = 240, 239
NOALLOC_MAX_SAMPLES = 1 << 21
NOALLOC_SCRATCH_BYTES = 42 << 20


class SoraError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().sora_adsb_strerror(code).decode()
        super().__init__(f"sora_adsb error {code} ({msg})" + (f": {detail}" if detail else ""))


class Frame(C.Structure):
    _fields_ = [
        ("position", C.c_uint64),
        ("bits", C.c_uint32 * 4),
        ("amplitude_56", C.c_double),
        ("syndrome", C.c_uint32),
        ("address", C.c_uint32),
        ("sum56", C.c_uint32),
        ("df", C.c_uint8),
        ("corrected_bit", C.c_int8),
        ("flags", C.c_uint8),
        ("reserved", C.c_uint8),
    ]


FRAME_DTYPE = np.dtype([
    ("position", "<u8"), ("bits", "<u4", (4,)), ("amplitude_56", "<f8"),
    ("syndrome", "<u4"), ("address", "<u4"), ("sum56", "<u4"),
    ("df", "u1"), ("corrected_bit", "i1"), ("flags", "u1"), ("reserved", "u1"),
])
assert FRAME_DTYPE.itemsize == C.sizeof(Frame) == 48

# struct sora_adsb_wire_frame (SORA_OPT_WIRE_FRAMES)
WIRE_DTYPE = np.dtype([
    ("bits", "<u4", (3,)), ("bits3_cb_flags", "<u4"), ("position_lo", "<u4"),
    ("pos_hi_amp_hi", "<u4"), ("amp_lo", "<u4"),
])
assert WIRE_DTYPE.itemsize == 28


class Config(C.Structure):
    _fields_ = [
        ("max_chunk_samples", C.c_uint64),
        ("max_candidates", C.c_uint64),
        ("max_squitters", C.c_uint64),
        ("max_frames", C.c_uint64),
        ("emit_mode", C.c_int32),
        ("options", C.c_int32),
        ("stream", C.c_void_p),
    ]


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("positions", "candidates", "squitters", "dirty_squitters",
                 "replies", "frames", "corrected")]


class SynthFrame(C.Structure):
    _fields_ = [
        ("position", C.c_uint64),
        ("bits", C.c_uint32 * 4),
        ("amp_i", C.c_int16),
        ("amp_q", C.c_int16),
        ("nbits", C.c_uint16),
        ("kind", C.c_uint16),
    ]


SYNTH_FRAME_DTYPE = np.dtype([
    ("position", "<u8"), ("bits", "<u4", (4,)), ("amp_i", "<i2"), ("amp_q", "<i2"),
    ("nbits", "<u2"), ("kind", "<u2"),
])
assert SYNTH_FRAME_DTYPE.itemsize == C.sizeof(SynthFrame) == 32


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("n_samples", C.c_uint64),
        ("encoding", C.c_int32),
        ("n_icao", C.c_uint32),
        ("mean_gap", C.c_uint32),
        ("df11_permille", C.c_uint32),
        ("reply_permille", C.c_uint32),
        ("unknown_permille", C.c_uint32),
        ("flip_permille", C.c_uint32),
        ("amp_min", C.c_int32),
        ("amp_max", C.c_int32),
        ("amp_log_uniform", C.c_int32),
        ("first_position", C.c_uint32),
    ]


READ_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t)
END = 1

# every symbol include/sora_adsb_gpu.h and include/sora_synth.h declare
_SIGNATURES = {
    "sora_adsb_version": (C.c_char_p, []),
    "sora_adsb_strerror": (C.c_char_p, [C.c_int]),
    "sora_adsb_device_count": (C.c_int, []),
    "sora_adsb_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Config)]),
    "sora_adsb_destroy": (None, [C.c_void_p]),
    "sora_adsb_last_error": (C.c_char_p, [C.c_void_p]),
    "sora_adsb_reset": (C.c_int, [C.c_void_p]),
    "sora_adsb_scan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                 C.c_uint64, C.c_uint64]),
    "sora_adsb_squitter_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "sora_adsb_copy_squitters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sora_adsb_add_known": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sora_adsb_host_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sora_adsb_export_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sora_adsb_add_known_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                             C.c_int]),
    "sora_adsb_resolve": (C.c_int, [C.c_void_p]),
    "sora_adsb_submit_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                          C.c_uint64, C.c_uint64]),
    "sora_adsb_submit_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                        C.c_uint64, C.c_uint64]),
    "sora_adsb_stream_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, READ_FN,
                                         C.c_void_p]),
    "sora_adsb_stream_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_size_t)]),
    "sora_adsb_stream_end": (C.c_int, [C.c_void_p]),
    "sora_adsb_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
    "sora_adsb_collect_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]),
    "sora_adsb_collect_wire": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t)]),
    "sora_adsb_collect_wire_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_size_t)]),
    "sora_adsb_unpack_frames": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "sora_adsb_get_counts": (C.c_int, [C.c_void_p, C.POINTER(Counts)]),
    "sora_adsb_list_regrows": (C.c_int, [C.c_void_p]),
    "sora_adsb_synchronize": (C.c_int, [C.c_void_p]),
    "sora_adsb_set_resolve_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sora_adsb_last_scan_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sora_adsb_scan_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "sora_adsb_decode_noalloc": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sora_adsb_magnitude": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                      C.c_void_p]),
    "sora_adsb_device_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "sora_adsb_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "sora_adsb_copy_to_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sora_adsb_copy_to_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sora_adsb_host_register": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t]),
    "sora_adsb_host_unregister": (C.c_int, [C.c_int, C.c_void_p]),
    "sora_synth_default_params": (C.c_int, [C.c_int, C.POINTER(SynthParams)]),
    "sora_synth_make_frames": (C.c_size_t, [C.POINTER(SynthParams), C.c_void_p,
                                            C.c_size_t]),
    "sora_synth_fill_host": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                       C.c_uint64, C.c_void_p, C.c_size_t]),
    "sora_synth_fill_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                         C.c_uint64, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "sora_modes_crc24": (C.c_uint32, [C.POINTER(C.c_uint32), C.c_int]),
    "sora_modes_build_df17": (None, [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                     C.c_uint64]),
    "sora_modes_build_df11": (None, [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]),
    "sora_modes_build_short_reply": (None, [C.POINTER(C.c_uint32), C.c_uint32,
                                            C.c_uint32, C.c_uint32]),
    "sora_modes_build_long_reply": (None, [C.POINTER(C.c_uint32), C.c_uint32,
                                           C.c_uint32, C.c_uint64, C.c_uint32]),
}

_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C sora_amd/csrc` (there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64
        # (same SONAME as /opt/rocm's).  Load torch's first when torch is
        # there, so that this library binds to the runtime torch.cuda and RCCL
        # use; two runtimes in one process cannot both open the GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def hip_runtime_paths():
    """The libamdhip64 / libhsa-runtime64 files mapped into this process."""
    out = set()
    with open("/proc/self/maps") as fh:
        for line in fh:
            if "libamdhip64" in line or "libhsa-runtime64" in line:
                out.add(line.split()[-1])
    return sorted(out)


def exported_symbols():
    return sorted(_SIGNATURES)


def check(code, ctx=None):
    if code != OK:
        detail = ""
        if ctx:
            detail = lib().sora_adsb_last_error(ctx).decode(errors="replace")
        raise SoraError(code, detail)


def bits_to_string(bits4):
    """112 '0'/'1' characters from the MSB-first word quadruple of a frame."""
    v = (int(bits4[0]) << 96) | (int(bits4[1]) << 64) | (int(bits4[2]) << 32) | int(bits4[3])
    return format(v >> 16, "0112b")


def string_to_bits(s):
    v = int(s.ljust(112, "0")[:112], 2) << 16
    return [(v >> 96) & 0xFFFFFFFF, (v >> 64) & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF,
            v & 0xFFFFFFFF]
