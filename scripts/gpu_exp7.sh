#!/bin/bash
O=gpurun_out; C=pixparse_amd/csrc
for cfg in "-DG_TIMING=0 -DG_CU_STAGGER=0" "-DG_TIMING=0 -DG_CU_STAGGER=1" "-DG_TIMING=8 -DG_CU_STAGGER=1"; do
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $cfg -c $C/gemm256.hip -o $C/gemm256.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
echo "######## $cfg"
python scripts/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | grep -E "==|tile [0-9]+:" | grep -v "tile [3-9]:\|tile 1[01]:"
done | tee $O/e7_timeline_stagger.log
