#!/bin/bash
# mid-round check: the whole -m gpu suite on the current library + the per-launch-shape trace of the default step
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6_mid_pytest.txt
cat gpurun_out/r6_mid_pytest.txt
bash scripts/gpu_trace_shapes.sh r6_mid > /dev/null 2>&1
head -70 gpurun_out/r6_mid_by_shape.txt
