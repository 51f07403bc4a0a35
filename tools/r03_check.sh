#!/bin/bash
# GPU tests, the resolve alone, the default bench and the events schedule: the
# quick check of a kernel change (tools/round_profile.sh is the full set).
set -e -o pipefail
O=gpurun_out/${1:-r03_check}
mkdir -p "$O"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread \
    > "$O/pytest_gpu.log" 2>&1 || { tail -40 "$O/pytest_gpu.log"; exit 1; }
tail -2 "$O/pytest_gpu.log"
timeout -k 10 120 python tools/resolve_only.py "" | tee "$O/resolve.txt"
timeout -k 10 120 python tools/resolve_only.py "" 9e8 | tee -a "$O/resolve.txt"
run() { local name=$1; shift
    timeout -k 10 200 python bench.py --no-cpu "$@" > "$O/$name.json" 2> "$O/$name.err" || { tail -5 "$O/$name.err"; exit 1; }
    python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_unshared'))"
}
run full_two
run full_events --schedule events
run shard_two --samples 9e8 --steps 100
run shardc_two --samples 9e8 --steps 100 --force-collective
run sc16_two --encoding sc16 --samples 3.6e9
