#!/bin/bash
# GPU contact of the forward stream: parity tests, then timing against the compiler-scheduled kernel
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "attention" 2>&1 | tail -15 > gpurun_out/f4w_tests.txt
cat gpurun_out/f4w_tests.txt
timeout 300 python scripts/bench_attn_fwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/f4w_bench.txt
