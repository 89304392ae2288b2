#!/bin/bash
# same-box A/B of the whole train step for run-time knobs (environment variables):  ab_env.sh "CRL_WGRAD_STREAM=0" "CRL_WGRAD_STREAM=1" ...
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
trap - EXIT      # nothing is rebuilt here
for rep in 1 2; do
for kv in "$@"; do
  echo "== $kv: $(env $kv python bench.py --no-cpu-baseline --no-roofline --no-host-leg --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step", "loss", d["loss"])')"
done; done
