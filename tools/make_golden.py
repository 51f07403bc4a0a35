#!/usr/bin/env python3
"""Generates tests/golden/ from the COMPILED REFERENCE (oracle/_ref).

Run in the build container, where /root/reference exists:
    make -C oracle ref && python tools/make_golden.py

Every fixture is data only: an input capture (raw IQ bytes) and what the
reference's own code printed for it -- the candidate stream with positions
(oracle/ref_candidates_driver.c), and the stdout of the reference main loop
with stream time (oracle/ref_shim_plane_iq.c) in both output modes.  Nothing
of the reference's source is stored.  The reference holds no fixtures of its
own, so these are the pinning vectors of the oracle.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob          # noqa: E402
from sora_amd import synth           # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

KLM = "8D4840D6202CC371C32CE0576098"     # 4840D6 KLM1023
DF11 = "5D4840D6F8740F"


def clean_capture(n, frames_at, high=250, low=128):
    """Noise-free uc8: Q = 128, I = low, pulses at `high`."""
    table = synth.frame_table([(p, synth.hex_to_bits(h), high - low, 0, 4 * len(h))
                               for p, h in frames_at])
    iq = np.full(2 * n, 128, dtype=np.uint8)
    iq[0::2] = low
    for f in table:
        for d in range(240):
            i = int(f["position"]) + d
            if i >= n:
                break
            hi = d in (0, 2, 7, 9)
            if d >= 16 and d - 16 < 2 * int(f["nbits"]):
                k = (d - 16) >> 1
                bit = (int(f["bits"][k >> 5]) >> (31 - (k & 31))) & 1
                hi = bool(bit ^ ((d - 16) & 1))
            if hi:
                iq[2 * i] = high
    return iq


def cases():
    rng = np.random.default_rng(20240611)
    out = {}
    out["klm_n1240"] = clean_capture(1240, [(1000, KLM)])
    out["klm_n1239"] = clean_capture(1239, [(1000, KLM)])
    out["df11_df17"] = clean_capture(4000, [(500, DF11), (1500, KLM), (2500, DF11)])
    out["uniform_noise"] = rng.integers(0, 256, size=2 * 65536, dtype=np.uint8)
    # odd byte count: the trailing byte is dropped (plane-iq.c:131, :138)
    out["uniform_noise_odd"] = rng.integers(0, 256, size=2 * 5000 + 1, dtype=np.uint8)
    p = synth.default_params(2, n_samples=100000, mean_gap=700, reply_permille=250,
                             unknown_permille=400, df11_permille=300, n_icao=6,
                             amp_min=10, amp_max=110, first_position=300)
    out["mixed_traffic"] = synth.fill_host(p.n_samples, p.encoding, p.seed,
                                           synth.make_frames(p))
    return out


def main():
    if not ob.have_reference():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` first")
    os.makedirs(GOLDEN, exist_ok=True)
    index = {}
    for name, iq in cases().items():
        raw = np.ascontiguousarray(iq, dtype=np.uint8).tobytes()
        with open(os.path.join(GOLDEN, name + ".uc8"), "wb") as fh:
            fh.write(raw)
        cands, final = ob.ref_candidates(raw, optimized=False)
        cands_o2, final_o2 = ob.ref_candidates(raw, optimized=True)
        assert cands == cands_o2 and final == final_o2, "reference -g and -O2 disagree"
        # position,112 bits as 28 hex digits,amplitude_56 (%a),amplitude (%a),accepted
        with open(os.path.join(GOLDEN, name + ".candidates.csv"), "w") as fh:
            for c in cands:
                fh.write("%d,%028X,%s,%s,%d\n" % (
                    c["position"], int(c["bits"], 2), float(c["amplitude_56"]).hex(),
                    float(c["amplitude"]).hex(), int(c["accepted"])))
        env = {"TZ": "UTC", "SORA_T0": "1700000000", "SORA_RATE": "2000000"}
        with open(os.path.join(GOLDEN, name + ".decoded.txt"), "w") as fh:
            fh.write(ob.ref_main_loop(raw, bitstring=False, shim=True, env=env))
        with open(os.path.join(GOLDEN, name + ".bitstring.txt"), "w") as fh:
            fh.write(ob.ref_main_loop(raw, bitstring=True, shim=True, env=env))
        # the ASCII input mode (planes-main-loop.c:15-47) fed with that bit-string text
        with open(os.path.join(GOLDEN, name + ".bitstring.txt"), "rb") as fh:
            ascii_in = fh.read()
        with open(os.path.join(GOLDEN, name + ".decoded_from_ascii.txt"), "w") as fh:
            fh.write(ob.ref_main_loop(ascii_in, bitstring=False, shim=False, env=env,
                                      from_raw=False))
        index[name] = dict(nbytes=len(raw), candidates=len(cands),
                           accepted=sum(c["accepted"] for c in cands),
                           final_position=final)
        print(name, index[name])
    with open(os.path.join(GOLDEN, "INDEX.json"), "w") as fh:
        json.dump(index, fh, indent=1, sort_keys=True)
    radio_file_fixture()


def radio_file_fixture():
    """Row a14: what the reference's radio->m->read() returns for one ragged file
    under every encoding of radio-file.h:6-12 (hex floats, one sample a line)."""
    import subprocess
    rng = np.random.default_rng(7)
    raw = rng.integers(0, 256, size=4 * 1000 + 3, dtype=np.uint8).tobytes()
    path = os.path.join(GOLDEN, "radio_file.bin")
    with open(path, "wb") as fh:
        fh.write(raw)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_radio_file")
    for enc in (1, 2, 3, 32, 33):
        p = subprocess.run([exe, path, str(enc), "300"], capture_output=True, check=True)
        with open(os.path.join(GOLDEN, f"radio_file.enc{enc}.txt"), "wb") as fh:
            fh.write(p.stdout)
        print("radio_file enc", enc, p.stdout.count(b"\n"), "samples")


if __name__ == "__main__":
    main()
