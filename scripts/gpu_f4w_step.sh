#!/bin/bash
# the forward stream in the step: attention tests, then the cfg-3 step with the forward in its three forms, alternating on one box
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "attention" 2>&1 | tail -4 > gpurun_out/f4w_tests.txt
cat gpurun_out/f4w_tests.txt
out=gpurun_out/r5_attn_fwd_step_ab.txt
: > $out
for rep in 1 2; do
for kv in "PIXPARSE_AMD_ATTN_FWD_MODE=1" "PIXPARSE_AMD_ATTN_FWD_MODE=3" "PIXPARSE_AMD_ATTN_FWD_MODE=0"; do
  echo "== $kv: $(env $kv python bench.py --no-cpu-baseline --no-roofline --no-host-leg --no-peak --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step", "non-attention", d.get("non_attention_ms_per_step"), "loss", d["loss"])')" >> $out
done; done
cat $out
