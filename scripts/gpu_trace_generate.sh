# rocprofv3 kernel trace of greedy generation (eager steps + graph replay) at the cfg-3 shapes -> gpurun_out/<tag>_generate_kernels.txt
tag=${1:-r3}
mkdir -p gpurun_out
python scripts/bench_generate.py --steps 64 2>&1 | grep -v Warning | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gen -- python3 $GRAFT_REPO_ROOT/scripts/bench_generate.py --steps 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_gen.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_gen -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_generate_kernels.txt <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 3 * 64        # warm-up run + eager run + graph run
print('per decode step (3 x 64 steps in the trace):')
for r in rows[:16]:
    n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:70]
    print(f"{n:70s} calls/step {int(r['Calls']) / steps:7.2f}  us/call {float(r['AverageNs']) / 1e3:8.1f}  us/step {float(r['TotalDurationNs']) / steps / 1e3:8.1f}")
PY
rm -rf gpurun_out/prof_gen
cat gpurun_out/${tag}_generate_kernels.txt
