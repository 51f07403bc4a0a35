"""Synthetic captures (include/sora_synth.h): frame tables and fills."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BYTES_PER_SAMPLE, SYNTH_FRAME_DTYPE, SynthParams, check, lib)

KIND_DF17, KIND_DF11, KIND_REPLY_KNOWN, KIND_REPLY_UNKNOWN = 1, 2, 3, 4
KIND_FLIPPED = 0x100


def default_params(config, **overrides):
    p = SynthParams()
    check(lib().sora_synth_default_params(config, C.byref(p)))
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def make_frames(params):
    """Frame table (numpy, SYNTH_FRAME_DTYPE) of a capture."""
    n = lib().sora_synth_make_frames(C.byref(params), None, 0)
    out = np.zeros(n, dtype=SYNTH_FRAME_DTYPE)
    if n:
        got = lib().sora_synth_make_frames(C.byref(params), out.ctypes.data, n)
        assert got == n
    return out


def frame_table(entries):
    """Hand-made table: entries of (position, bits4 or '0101..', amp_i, amp_q, nbits)."""
    out = np.zeros(len(entries), dtype=SYNTH_FRAME_DTYPE)
    for i, (pos, bits, ai, aq, nbits) in enumerate(entries):
        if isinstance(bits, str):
            bits = _lib.string_to_bits(bits)
        out[i] = (pos, tuple(bits), ai, aq, nbits, 0)
    order = np.argsort(out["position"], kind="stable")
    return out[order]


def fill_host(n_samples, encoding, seed, frames, first=0):
    """Bytes of samples [first, first+n_samples) computed on the CPU."""
    out = np.zeros(n_samples * BYTES_PER_SAMPLE[encoding], dtype=np.uint8)
    frames = np.ascontiguousarray(frames)
    check(lib().sora_synth_fill_host(out.ctypes.data, first, n_samples, encoding, seed,
                                     frames.ctypes.data if frames.size else None,
                                     frames.size))
    return out


def fill_device(d_ptr, n_samples, encoding, seed, frames, first=0, stream=None):
    """Same bytes written by the gfx950 fill kernel into device memory."""
    frames = np.ascontiguousarray(frames)
    check(lib().sora_synth_fill_device(d_ptr, first, n_samples, encoding, seed,
                                       frames.ctypes.data if frames.size else None,
                                       frames.size, stream))


def hex_to_bits(hexstr):
    v = int(hexstr, 16) << (128 - 4 * len(hexstr))
    return [(v >> 96) & 0xFFFFFFFF, (v >> 64) & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF,
            v & 0xFFFFFFFF]


def crc24(bits4, covered):
    arr = (C.c_uint32 * 4)(*bits4)
    return lib().sora_modes_crc24(arr, covered)


def build_df17(icao, me56, ca=5):
    arr = (C.c_uint32 * 4)()
    lib().sora_modes_build_df17(arr, icao, ca, me56)
    return list(arr)


def build_df11(icao, ca=5):
    arr = (C.c_uint32 * 4)()
    lib().sora_modes_build_df11(arr, icao, ca)
    return list(arr)


def build_short_reply(df, payload27, icao):
    arr = (C.c_uint32 * 4)()
    lib().sora_modes_build_short_reply(arr, df, payload27, icao)
    return list(arr)


def build_long_reply(df, payload27, mb56, icao):
    arr = (C.c_uint32 * 4)()
    lib().sora_modes_build_long_reply(arr, df, payload27, mb56, icao)
    return list(arr)
