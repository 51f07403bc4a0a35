"""sora_amd -- MI355X (gfx950) implementation of Sora's ADS-B receive chain.

Thin Python plumbing over the C ABI of libsora_adsb_gpu.so; the kernels, the
context and the synthetic generator are native (sora_amd/csrc).
"""
from . import _lib
from ._lib import (ENC_SC16, ENC_SC8, ENC_UC8, EMIT_ACCEPTED, EMIT_CANDIDATES,
                   FRAME_ACCEPTED, FRAME_CORRECTED, FRAME_DTYPE, FRAME_LONG, FRAME_SQUITTER,
                   HALO, WINDOW, WIRE_DTYPE,
                   SoraError, bits_to_string, string_to_bits)
from .adsb import AdsbContext, DeviceBuffer, unpack_frames
from . import synth

__all__ = ["AdsbContext", "DeviceBuffer", "synth", "SoraError", "ENC_UC8", "ENC_SC8",
           "ENC_SC16", "EMIT_ACCEPTED", "EMIT_CANDIDATES", "FRAME_DTYPE",
           "FRAME_LONG", "FRAME_SQUITTER", "FRAME_ACCEPTED", "FRAME_CORRECTED", "HALO",
           "WINDOW", "WIRE_DTYPE", "unpack_frames",
           "bits_to_string", "string_to_bits"]
