import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT)
from sora_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(ROOT, "sora_amd", "lib", sys.argv[1])
import torch, sora_amd
from sora_amd import ENC_UC8, AdsbContext, synth
n = 7_200_000_000
p = synth.default_params(2, n_samples=n); fr = synth.make_frames(p)
d = torch.empty(2*n+16, dtype=torch.uint8, device="cuda")
synth.fill_device(d.data_ptr(), n, ENC_UC8, p.seed, fr)
mf = n // 1024 + 65536
ctx = AdsbContext(n, max_squitters=mf, max_frames=mf, wire_frames=True)
ctx.scan(d, 2*n, ENC_UC8); ctx.synchronize()
best = 1e9
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.resolve(); ctx.synchronize()
    best = min(best, time.perf_counter() - t0)
print(sys.argv[1:] , "resolve ms", best*1e3, "frames", ctx.frame_count())
