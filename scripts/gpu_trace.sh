# rocprofv3 kernel trace of the default bench (cfg-3) -> gpurun_out/<tag>_kernel_stats.csv + the bench line of the same run
tag=${1:-r3}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-peak > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.err
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/${tag}_bench_under_rocprof.json | cut -c1-400
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
python3 - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/${tag}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/4e6:9.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print('total per step (4 steps incl. warmup)', tot/4e6)
PY
