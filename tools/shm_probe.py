#!/usr/bin/env python3
"""Diagnostic: can a /dev/shm mapping be page-locked on this box, and what does
a device-to-host copy into it achieve?"""
import ctypes as C, mmap, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sora_amd import _lib
L = _lib.lib()
n = 256 << 20
d = torch.ones(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

def rate(ptr, tag):
    for rep in range(3):
        t0 = time.perf_counter()
        L.sora_adsb_copy_to_host(0, ptr, d.data_ptr(), n)
        dt = time.perf_counter() - t0
    print(tag, "D2H %.1f GB/s" % (n / dt / 1e9), flush=True)

path = f"/dev/shm/sora_probe_{os.getpid()}"
with open(path, "wb") as fh:
    fh.truncate(n)
mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(n,))
os.unlink(path)
rate(mm.ctypes.data, "shm unregistered, untouched")
rc = L.sora_adsb_host_register(0, mm.ctypes.data, n)
print("register shm (touched by the copy) rc", rc, flush=True)
if rc == 0:
    rate(mm.ctypes.data, "shm registered")
    L.sora_adsb_host_unregister(0, mm.ctypes.data)
an = mmap.mmap(-1, n)
a = np.frombuffer(an, dtype=np.uint8)
rc = L.sora_adsb_host_register(0, a.ctypes.data, n)
print("register anonymous shared rc", rc, flush=True)
if rc == 0:
    rate(a.ctypes.data, "anon registered")
    L.sora_adsb_host_unregister(0, a.ctypes.data)
pv = np.zeros(n, dtype=np.uint8)
rc = L.sora_adsb_host_register(0, pv.ctypes.data, n)
print("register private heap rc", rc, flush=True)
if rc == 0:
    rate(pv.ctypes.data, "heap registered")
    L.sora_adsb_host_unregister(0, pv.ctypes.data)
# pinned staging + parallel CPU copy into the shared segment
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
shm_t = torch.from_numpy(mm)
for rep in range(3):
    t0 = time.perf_counter()
    pin.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t1 = time.perf_counter()
    shm_t.copy_(pin)
    t2 = time.perf_counter()
print("pinned D2H %.1f GB/s, then CPU copy into shm %.1f GB/s (%d threads)" %
      (n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9, torch.get_num_threads()))
