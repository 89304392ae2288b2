#!/bin/bash
# same-box A/B: the step with the live per-launch HIP events of crl_prof_* (roofline on) and without (--no-roofline)
O=gpurun_out
for i in 1 2; do
for f in "--no-roofline" ""; do
python bench.py --no-cpu-baseline --no-host-leg --steps 8 --warmup 2 $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags[$f]', d['value'], 'docs/s', d['ms_per_step'], 'ms/step')"
done; done | tee $O/prof_overhead.log
