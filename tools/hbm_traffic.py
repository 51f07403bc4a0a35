#!/usr/bin/env python3
"""HBM traffic of the scan kernel from two rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d F -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d W -- python3 bench.py --steps 2 --warmup 1 --no-cpu
    python tools/hbm_traffic.py F W uc8 bytes_per_gpu > profiles/hbm_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM) says:
both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
wide (16 B per lane) coalesced streaming reads, which is what the scan kernel
issues, so it is doubled; WRITE_SIZE is taken as is.  bench.py copies the
per-launch figure of the matching workload into roofline.traffic.
"""
import csv
import glob
import json
import os
import sys


def mean_counter(directory, counter, kernel):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"),
                          recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
    if not vals:
        sys.exit(f"no {counter} rows for {kernel} under {directory}")
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, enc, nbytes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    kernel = sys.argv[5] if len(sys.argv) > 5 else "adsb_scan_kernel"
    fetch, nf = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
    write, nw = mean_counter(write_dir, "WRITE_SIZE", kernel)
    out = {
        "kernel": kernel, "algorithmic_bytes": nbytes,
        "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
        "launches": [nf, nw],
        "read_bytes": 2 * fetch * 1024, "write_bytes": write * 1024,
        "traffic_bytes": 2 * fetch * 1024 + write * 1024,
        "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane streaming reads), KiB units",
    }
    print(json.dumps({f"{kernel}:{enc}:{nbytes}": out}, indent=1))


if __name__ == "__main__":
    main()
