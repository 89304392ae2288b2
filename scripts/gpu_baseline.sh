#!/bin/bash
# One GPU-box call: full -m gpu suite, default bench line, rocprofv3 kernel stats of the bench, FETCH/WRITE PMC passes.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_baseline.sh TAG'
TAG=${1:-r2}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/${TAG}_tests.log
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof.err
find $O/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg > $O/${TAG}_pmc_write.log 2>&1
cd $R
python scripts/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt 2>&1
# keep the merge-back small: drop the raw traces
find $O/${TAG}_prof $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -name '*kernel_trace.csv' -delete
find $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -name '*counter_collection.csv' -size +20M -delete
tail -3 $O/${TAG}_tests.log; cat $O/${TAG}_bench.json | cut -c1-400
