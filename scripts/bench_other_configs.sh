#!/bin/bash
# the other BASELINE.json configs through the same bench (no roofline / cpu legs): docs/s and ms/step
for spec in "cruller_small 8" "cruller_base 8" "cruller_base_960x640 8" "cruller_large_6layers 2"; do
  set -- $spec
  python bench.py --model $1 --batch $2 --no-cpu-baseline --no-roofline --no-host-leg --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 batch $2:', d['value'], 'docs/s', d['ms_per_step'], 'ms/step', 'step_mfma_frac', d['step_mfma_frac'], 'act GB', d['activation_gb'])"
done
