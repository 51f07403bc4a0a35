#!/bin/bash
# The measurements a kernel change is judged by, in one GPU call:
#   tools/round_profile.sh TAG [quick]
# writes under gpurun_out/TAG: the GPU tests' log, the default bench line, the
# rocprofv3 kernel statistics of the same workload with one capture in flight,
# and (unless "quick") the PMC passes for HBM traffic and VALU issue, one pass
# per run as the pool requires.  Every GPU step has a time limit of its own and
# the script stops at the first failure.
set -e -o pipefail
TAG=${1:?tag}
O=gpurun_out/$TAG
mkdir -p "$O"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp

timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread \
    > "$O/pytest_gpu.log" 2>&1 || { tail -30 "$O/pytest_gpu.log"; exit 1; }
tail -1 "$O/pytest_gpu.log"
timeout -k 10 300 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
cat "$O/bench_default.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- \
    python3 bench.py --steps 30 --warmup 5 --no-cpu --inflight 1 \
    > "$O/bench_inflight1_under_rocprof.json" 2> "$O/prof.err"
cp "$O"/prof/*/*_kernel_stats.csv "$O/inflight1_kernel_stats.csv"
head -6 "$O/inflight1_kernel_stats.csv"
[ "$2" = quick ] && exit 0

B="python3 bench.py --steps 2 --warmup 1 --no-cpu --inflight 1"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $B > /dev/null 2> "$O/pmc_fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $B > /dev/null 2> "$O/pmc_write.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d "$O/pmc_sq" -- $B > /dev/null 2> "$O/pmc_sq.err"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$O/pmc_grbm" -- $B > /dev/null 2> "$O/pmc_grbm.err"
KERNEL=$(python3 -c "import json,sys; print(json.load(open('$O/bench_default.json'))['roofline']['kernel'].split(' ')[0])")
python3 tools/hbm_traffic.py "$O/pmc_fetch" "$O/pmc_write" uc8 14400000000 "$KERNEL" > "$O/hbm_traffic.json"
cat "$O/hbm_traffic.json"
for d in pmc_fetch pmc_write pmc_sq pmc_grbm; do
    cat "$O"/$d/*/*counter_collection.csv | grep -E "Counter_Name|adsb_scan" > "$O/$d.csv" || true
done
