#!/bin/bash
O=gpurun_out; C=pixparse_amd/csrc; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/e8_tests.log 2>&1; tail -3 $O/e8_tests.log
python scripts/bench_kernels.py gemm2x 2>&1 | grep -v amdgpu | grep "pol=0" > $O/e8_kernels.log
python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 2>/dev/null | tail -1 | cut -c1-200
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DG_TIMING=0 -c $C/gemm256.hip -o $C/gemm256.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
python scripts/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | grep -E "==|tile [012]:" > $O/e8_timeline.log
cat $O/e8_timeline.log
