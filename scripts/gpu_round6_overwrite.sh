#!/bin/bash
# round 6: weight gradients overwrite the gradient arena on the first micro-step (PIXPARSE_AMD_WGRAD_OVERWRITE=0|1): model tests + step A/B on one box
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_00_dist_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r6_overwrite_pytest.txt
cat gpurun_out/r6_overwrite_pytest.txt
: > gpurun_out/r6_wgrad_overwrite_step_ab.txt
for rep in 1 2 3; do
  for t in 0 1; do
    echo "== overwrite=$t: $(PIXPARSE_AMD_WGRAD_OVERWRITE=$t python bench.py --no-cpu-baseline --no-roofline --no-host-leg --no-peak --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step loss", d["loss"])')" >> gpurun_out/r6_wgrad_overwrite_step_ab.txt
  done
done
cat gpurun_out/r6_wgrad_overwrite_step_ab.txt
