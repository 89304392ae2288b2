#!/bin/bash
# HBM traffic per kernel launch: two rocprofv3 PMC passes over one cfg-3 step (FETCH_SIZE and WRITE_SIZE cannot share a pass), then
# scripts/pmc_traffic.py -> gpurun_out/r3_pmc_traffic.json (copy to profiles/).  Program directly behind `--`, no trace domains beside --pmc.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r3_pmc_traffic.json | head -30
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
# cfg-1 (cruller_small, batch 2): where do its 7 ms go?
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -- python3 $R/bench.py --model cruller_small --batch 2 --steps 5 --warmup 2 --graph-step off --no-cpu-baseline --no-roofline --no-host-leg > $R/gpurun_out/prof_small.log 2>&1
cd $R
find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_cfg1_kernel_stats.csv
rm -rf gpurun_out/prof_small
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3_cfg1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('cfg-1: total kernel time per step %.3f ms, %d launches per step' % (tot/7/1e6, calls/7))
for r in rows[:22]:
    print('%-90s calls/step %6.1f  us/call %8.1f  ms/step %6.3f' % (r['Name'][:90], int(r['Calls'])/7, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/7/1e6))
PY
