#!/bin/bash
# after a change to attention.hip: the GPU suite, the two HBM-traffic PMC passes (bench.py drops roofline.traffic when the kernel source is
# newer than profiles/r4_pmc_traffic.json), then the default bench line with the fresh figure
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r4_pmc_traffic.json | head -4
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp gpurun_out/r4_pmc_traffic.json profiles/r4_pmc_traffic.json
python bench.py > gpurun_out/r4_bench_default_output.json 2> gpurun_out/r4_bench_default_output.err
python -c "
import json; d=json.load(open('gpurun_out/r4_bench_default_output.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), r.get('traffic_algorithmic'), r.get('attn_bwd_chain'), r['attention_bwd_total'], d['non_attention_ms_per_step'], d['cpu_baseline'])"
