#!/bin/bash
O=gpurun_out; C=pixparse_amd/csrc
for cap in 256 32; do
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DG_TIMING=0 -DG_GRID_CAP=$cap -c $C/gemm256.hip -o $C/gemm256.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
echo "######## grid cap $cap"
TL_M=$((cap*256*3)) python scripts/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | grep -E "==|tile [012]:"
done | tee $O/e3_timeline_cap.log
