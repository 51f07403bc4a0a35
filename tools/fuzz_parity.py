#!/usr/bin/env python3
"""Randomized parity sweep (not part of the test tiers): captures of random
length, traffic density, encoding, emit mode and scan geometry through the C ABI
against the oracle.  python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                   # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

import oracle_binding as ob                          # noqa: E402
from conftest import assert_same_frames, frames_as_dicts   # noqa: E402
import sora_amd                                      # noqa: E402
from sora_amd import (EMIT_ACCEPTED, EMIT_CANDIDATES, ENC_SC8, ENC_SC16, ENC_UC8,  # noqa: E402
                      AdsbContext, synth)

dense7 = np.array([9, 1, 8, 1, 2, 2, 2], dtype=np.int64)
bad = 0
for case in range(cases):
    enc = [ENC_UC8, ENC_UC8, ENC_SC8, ENC_SC16][rng.integers(0, 4)]
    n = int(rng.choice([rng.integers(240, 3000), rng.integers(3000, 200000),
                        rng.integers(200000, 3000000)]))
    gap = int(rng.choice([60, 300, 1500, 6000]))
    p = synth.default_params(4 if enc == ENC_SC16 else 2, n_samples=n, mean_gap=gap,
                             reply_permille=int(rng.integers(0, 500)),
                             unknown_permille=int(rng.integers(0, 500)),
                             n_icao=int(rng.integers(1, 40)), seed=int(rng.integers(1, 1 << 30)))
    p.encoding = enc
    raw = synth.fill_host(p.n_samples, enc, p.seed, synth.make_frames(p)).tobytes()
    if enc != ENC_SC16 and rng.integers(0, 4) == 0:   # a dense stretch somewhere
        m = int(rng.integers(1000, 40000))
        iq = np.full(2 * m, 128, dtype=np.uint8)
        iq[0::2] = 128 + 10 * dense7[np.arange(m) % 7]
        if enc == ENC_SC8:
            iq = (iq.astype(np.int16) - 128).astype(np.int8).view(np.uint8)
        at = int(rng.integers(0, n)) * 2
        raw = raw[:at] + iq.tobytes() + raw[at:]
        n = len(raw) // 2
    os.environ["SORA_ADSB_SCAN_ROUNDS"] = str(int(rng.choice([1, 2, 16, 32, 64])))
    os.environ["SORA_ADSB_SCAN_MIN_TILES"] = str(int(rng.choice([1, 2, 8, 50])))
    emit = [EMIT_ACCEPTED, EMIT_CANDIDATES][rng.integers(0, 2)]
    small = rng.integers(0, 3) == 0                   # lists that have to grow
    kw = dict(max_candidates=64, max_squitters=64, max_frames=64) if small else \
        dict(max_candidates=n // 2 + 1024, max_squitters=n // 2 + 1024)
    want = ob.scan(raw, enc)
    if emit == EMIT_ACCEPTED:
        want = [c for c in want if c["accepted"]]
    try:
        with AdsbContext(max(n, 1), emit_mode=emit, **kw) as ctx:
            ctx.submit_host(raw, enc)
            got = frames_as_dicts(ctx.collect())
        assert_same_frames(got, want)
    except Exception as e:                            # noqa: BLE001
        bad += 1
        print(f"case {case}: enc {enc} n {n} gap {gap} emit {emit} small {small} "
              f"rounds {os.environ['SORA_ADSB_SCAN_ROUNDS']} "
              f"min_tiles {os.environ['SORA_ADSB_SCAN_MIN_TILES']} seed {p.seed}: "
              f"{type(e).__name__}: {str(e)[:300]}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} bad", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
