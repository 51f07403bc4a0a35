#!/bin/bash
# Samples power and clocks (rocm-smi) while the default bench workload runs:
#   tools/power_watch.sh [bench args...]   -> gpurun_out/power_watch.txt
# One bench process in the background, a few rocm-smi samples beside it.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/power_watch.txt
: > "$O"
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" >> "$O"
echo "---- under load ----" >> "$O"
timeout -k 10 120 python bench.py --steps 1500 --warmup 5 --no-cpu "$@" > gpurun_out/power_watch_bench.json 2>/dev/null &
B=$!
sleep 14
for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk|socclk" >> "$O"
    echo "--" >> "$O"
    sleep 0.4
done
wait $B
echo "bench exit $?" >> "$O"
cat gpurun_out/power_watch_bench.json >> "$O"
