#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace (+ --memory-copy-trace)
CSV directory: for the last few steps, every kernel/copy with its start offset,
duration and the gap to the previous one on the device.
    python tools/trace_timeline.py gpurun_out/xyz [n_kernels]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                     .split("(")[0][:48], r.get("Stream_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     "copy " + r.get("Direction", ""), r.get("Stream_Id", "")))
rows.sort()
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[-n - skip:len(rows) - skip]
t0 = rows[0][0]
busy_end = rows[0][0]
for s, e, name, stream in rows:
    gap = s - busy_end
    print(f"{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  gap {gap / 1e3:8.1f}  q{stream:>3s} {name}")
    busy_end = max(busy_end, e)
