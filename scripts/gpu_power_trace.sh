#!/bin/bash
# sample power / clocks with rocm-smi while bench.py runs (is the step power-limited?)
O=gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > $O/power_idle.txt
( for i in $(seq 1 60); do rocm-smi -P -c 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/power_trace.txt &
SMI=$!
python bench.py --no-cpu-baseline --no-host-leg --steps 40 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
kill $SMI 2>/dev/null
cat $O/power_idle.txt | head -30
tail -25 $O/power_trace.txt
