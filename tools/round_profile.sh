#!/bin/bash
# The measurements a kernel change is judged by, in one GPU call:
#   tools/round_profile.sh TAG [quick]
# writes under gpurun_out/TAG: the GPU tests' log, the default bench line (two
# captures in flight, two-stream schedule) and the driver's 20-step form of it,
# the rocprofv3 kernel statistics of that same default command AND of the
# events schedule with one capture in flight (every kernel with the GPU to
# itself), the side measurements (900 M-sample shard with and without the N>1
# code path, dense sc16, archive), and (unless "quick") the PMC passes for HBM
# traffic and VALU issue, one pass per run as the pool requires.  Every GPU step
# has a time limit of its own and the script stops at the first failure.
set -e -o pipefail
TAG=${1:?tag}
O=gpurun_out/$TAG
mkdir -p "$O"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp

timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread \
    > "$O/pytest_gpu.log" 2>&1 || { tail -30 "$O/pytest_gpu.log"; exit 1; }
tail -1 "$O/pytest_gpu.log"
timeout -k 10 300 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
cat "$O/bench_default.json"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > "$O/bench_20steps.json" 2> "$O/bench_20steps.err"
side() { local name=$1; shift
    timeout -k 10 300 python bench.py --no-cpu "$@" > "$O/$name.json" 2> "$O/$name.err"
    python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_unshared'))"
}
side bench_events --schedule events
side bench_full_frames --frames full
side shard900M --samples 9e8 --steps 100
side shard900M_force_collective --samples 9e8 --steps 100 --force-collective
side shard900M_events --samples 9e8 --steps 100 --schedule events
side shard900M_force_collective_events --samples 9e8 --steps 100 --force-collective --schedule events
side sc16_dense --encoding sc16 --samples 3.6e9
side archive8 --archive 8
# where a short shard's step goes on the N>1 path and locally (HIP events)
timeout -k 10 200 python tools/shard_timeline.py 9e8 collective 6 300 > "$O/shard900M_timeline_collective.txt" 2> /dev/null
timeout -k 10 200 python tools/shard_timeline.py 9e8 local 6 300 > "$O/shard900M_timeline_local.txt" 2> /dev/null
tail -1 "$O/shard900M_timeline_collective.txt"
# the same default command under rocprofv3 (what roofline.kernel_ms must agree with)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof2" -- \
    python3 bench.py --steps 30 --warmup 5 --no-cpu \
    > "$O/bench_default_under_rocprof.json" 2> "$O/prof2.err"
cp "$O"/prof2/*/*_kernel_stats.csv "$O/default_kernel_stats.csv"; rm -rf "$O/prof2"
head -7 "$O/default_kernel_stats.csv" | cut -c1-160
# every kernel with the GPU to itself
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- \
    python3 bench.py --steps 30 --warmup 5 --no-cpu --schedule events --inflight 1 \
    > "$O/bench_inflight1_under_rocprof.json" 2> "$O/prof.err"
cp "$O"/prof/*/*_kernel_stats.csv "$O/inflight1_kernel_stats.csv"; rm -rf "$O/prof"
head -8 "$O/inflight1_kernel_stats.csv" | cut -c1-160
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof16" -- \
    python3 bench.py --steps 30 --warmup 5 --no-cpu --schedule events --inflight 1 --encoding sc16 --samples 3.6e9 \
    > "$O/bench_sc16_inflight1_under_rocprof.json" 2> "$O/prof16.err"
cp "$O"/prof16/*/*_kernel_stats.csv "$O/sc16_inflight1_kernel_stats.csv"; rm -rf "$O/prof16"
[ "$2" = quick ] && exit 0

B="python3 bench.py --steps 2 --warmup 1 --no-cpu --schedule events --inflight 1"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $B > /dev/null 2> "$O/pmc_fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $B > /dev/null 2> "$O/pmc_write.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d "$O/pmc_sq" -- $B > /dev/null 2> "$O/pmc_sq.err"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$O/pmc_grbm" -- $B > /dev/null 2> "$O/pmc_grbm.err"
KERNEL=$(python3 -c "import json,sys; print(json.load(open('$O/bench_default.json'))['roofline']['kernel'].split(' ')[0])")
python3 tools/hbm_traffic.py "$O/pmc_fetch" "$O/pmc_write" uc8 14400000000 "$KERNEL" > "$O/hbm_traffic.json"
cat "$O/hbm_traffic.json"
# the same two passes over the dense sc16 side workload
B16="$B --encoding sc16 --samples 3.6e9"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc16_fetch" -- $B16 > /dev/null 2> "$O/pmc16_fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc16_write" -- $B16 > /dev/null 2> "$O/pmc16_write.err"
python3 tools/hbm_traffic.py "$O/pmc16_fetch" "$O/pmc16_write" sc16 14400000000 "adsb_scan_kernel<true>" > "$O/hbm_traffic_sc16.json"
cat "$O/hbm_traffic_sc16.json"
for d in pmc_fetch pmc_write pmc_sq pmc_grbm pmc16_fetch pmc16_write; do
    cat "$O"/$d/*/*counter_collection.csv | grep -E "Counter_Name|adsb_scan|filter_kernel|finalize_kernel" > "$O/$d.csv" || true
    rm -rf "$O/$d"
done
