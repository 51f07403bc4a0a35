import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from sora_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join('/root/repo/sora_amd/lib', sys.argv[1])
import numpy as np, oracle_binding as ob, sora_amd
from conftest import frames_as_dicts
rng = np.random.default_rng(101)
raw = rng.integers(0, 256, size=2 * 300007, dtype=np.uint8).tobytes()
n = len(raw)//2
with sora_amd.AdsbContext(n, emit_mode=1, max_candidates=n//2+1024, max_squitters=n//2+1024) as ctx:
    ctx.submit_host(raw, 1)
    got = frames_as_dicts(ctx.collect())
want = ob.scan(raw)
print(sys.argv[1:], len(got), len(want), 'positions equal:', [g['position'] for g in got] == [w['position'] for w in want])
gp = set(g['position'] for g in got); wp = set(w['position'] for w in want)
print(' missing', len(wp-gp), 'extra', len(gp-wp), sorted(wp-gp)[:5], sorted(gp-wp)[:5])
