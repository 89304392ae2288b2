#!/bin/bash
# the other BASELINE.json configs through the same bench (no roofline / cpu legs): docs/s and ms/step, eager launches against the
# hipGraph replay of the micro-step (TaskCrullerPretrainCfg.graph_step; auto = on for models under 250 M parameters)
for spec in "cruller_small 2" "cruller_small 8" "cruller_base_960x640 8" "cruller_large_6layers 2"; do
  set -- $spec
  for g in off on; do
    python bench.py --model $1 --batch $2 --graph-step $g --no-cpu-baseline --no-roofline --no-host-leg --steps ${STEPS:-10} --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 batch $2 graph=$g:', d['value'], 'docs/s', d['ms_per_step'], 'ms/step', 'step_mfma_frac', d['step_mfma_frac'], 'loss', d['loss'], '|', d['launch'])"
  done
done
