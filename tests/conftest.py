import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TZ", "UTC")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """A fixture produced by tools/make_golden.py from the compiled reference."""
    with open(os.path.join(GOLDEN, name + ".uc8"), "rb") as fh:
        raw = fh.read()
    cands = []
    with open(os.path.join(GOLDEN, name + ".candidates.csv")) as fh:
        for line in fh:
            pos, bits, a56, amp, acc = line.strip().split(",")
            cands.append(dict(position=int(pos), bits=format(int(bits, 16), "0112b"),
                              amplitude_56=float.fromhex(a56),
                              amplitude=float.fromhex(amp), accepted=acc == "1"))
    text = {}
    for mode in ("decoded", "bitstring"):
        with open(os.path.join(GOLDEN, f"{name}.{mode}.txt")) as fh:
            text[mode] = fh.read()
    return raw, cands, text


GOLDEN_NAMES = ["klm_n1240", "klm_n1239", "df11_df17", "uniform_noise",
                "uniform_noise_odd", "mixed_traffic"]


@pytest.fixture(scope="session")
def emulator():
    """CPU lane emulator of the scan kernel (tests/emul/tile_emul.cpp)."""
    src = os.path.join(ROOT, "tests", "emul", "tile_emul.cpp")
    so = os.path.join(ROOT, "tests", "emul", "libtile_emul.so")
    hdr = os.path.join(ROOT, "sora_amd", "csrc", "adsb_tile.h")
    if (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
            or os.path.getmtime(so) < os.path.getmtime(hdr)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, src],
                       check=True)
    L = C.CDLL(so)
    L.emul_scan8.restype = C.c_size_t
    L.emul_scan8.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                             C.c_void_p, C.c_size_t]
    L.emul_scan16.restype = C.c_size_t
    L.emul_scan16.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                              C.c_void_p, C.c_size_t]
    return L


def frames_as_dicts(frames):
    """numpy FRAME_DTYPE array -> list of dicts comparable with the oracle's."""
    from sora_amd import FRAME_ACCEPTED, bits_to_string
    out = []
    for f in frames:
        out.append(dict(position=int(f["position"]), bits=bits_to_string(f["bits"]),
                        amplitude_56=float(f["amplitude_56"]),
                        amplitude=float(f["amplitude_56"]) / 56,
                        df=int(f["df"]), syndrome=int(f["syndrome"]),
                        address=int(f["address"]),
                        accepted=bool(int(f["flags"]) & FRAME_ACCEPTED),
                        corrected_bit=int(f["corrected_bit"])))
    return out


def assert_same_frames(got, want, fields=("position", "bits", "amplitude_56", "amplitude",
                                          "df", "syndrome", "address", "accepted",
                                          "corrected_bit")):
    assert len(got) == len(want), f"{len(got)} frames, expected {len(want)}"
    for i, (g, w) in enumerate(zip(got, want)):
        for k in fields:
            if k in w:
                assert g[k] == w[k], f"frame {i} at {w['position']}: {k} {g[k]!r} != {w[k]!r}"
