mkdir -p gpurun_out
(timeout 900 python scripts/bench_loader.py --docs 264 --workers 0,4,8,16,32,64 --fmt png 2>&1 | grep -v Warning) > gpurun_out/loader_png.txt
(timeout 600 python scripts/bench_loader.py --docs 264 --workers 8,32 --fmt jpg 2>&1 | grep -v Warning) > gpurun_out/loader_jpg.txt
(timeout 600 python scripts/bench_loader.py --docs 136 --workers 32 --fmt png --cpu-resize 2>&1 | grep -v Warning) > gpurun_out/loader_cpu_resize.txt
cat gpurun_out/loader_*.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-peak > $GRAFT_REPO_ROOT/gpurun_out/prof_r3a_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r3a.err
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/prof_r3a_bench.json | cut -c1-600
find gpurun_out/prof_r3a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3a_kernel_stats.csv
head -40 gpurun_out/r3a_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_r3a
