# per-(kernel, grid) launch durations of one bench run: gpurun_out/<tag>_by_shape.txt
tag=${1:-r3}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-peak --no-roofline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/${tag}_shapes_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.err
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/${tag}_shapes_bench.json | cut -c1-300
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_by_shape.txt <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    agg[(name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
nsteps = 4
tot = sum(sum(v) for v in agg.values())
print(f'total {tot / nsteps / 1e3:.2f} ms/step over {nsteps} steps')
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / nsteps < 50: break
    print(f'{k[0][:44]:44s} grid {k[1]:6d}x{k[2]:3d}x{k[3]:3d} launches/step {len(v) / nsteps:6.1f} avg {sum(v) / len(v):8.1f} us  min {min(v):8.1f}  ms/step {sum(v) / nsteps / 1e3:7.3f}')
PY
rm -rf gpurun_out/prof_$tag
cat gpurun_out/${tag}_by_shape.txt
