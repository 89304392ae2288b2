#!/bin/bash
O=gpurun_out; C=pixparse_amd/csrc; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/e4_tests.log 2>&1; tail -3 $O/e4_tests.log
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DG_TIMING=0 -c $C/gemm256.hip -o $C/gemm256.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
python scripts/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | grep -E "==|tile [012]:" > $O/e4_timeline.log
bash scripts/ab_kernels.sh gemm256.hip gemm2x "pol=0" "-DG_EPI_XPOSE=0" "-DG_EPI_XPOSE=1" > $O/e4_ab_kernels.log 2>&1
bash scripts/ab_flags.sh gemm256.hip "-DG_EPI_XPOSE=0" "-DG_EPI_XPOSE=1" "-DG_EPI_XPOSE=0" "-DG_EPI_XPOSE=1" > $O/e4_ab_step.log 2>&1
cat $O/e4_ab_step.log
