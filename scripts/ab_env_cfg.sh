#!/bin/bash
# same-box A/B of another config's train step for run-time knobs:  MODEL=cruller_base_960x640 BATCH=8 ab_env_cfg.sh "ENV=0" "ENV=1" ...
cd "$(dirname "$0")/.."
for rep in 1 2; do
for kv in "$@"; do
  echo "== $kv: $(env $kv python bench.py --model ${MODEL:-cruller_base_960x640} --batch ${BATCH:-8} --graph-step off --no-cpu-baseline --no-roofline --no-host-leg --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step", "frac", d["step_mfma_frac"], "loss", d["loss"])')"
done; done
