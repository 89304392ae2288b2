#!/bin/bash
# same-box A/B of the attention micro-benchmark for compile-time knobs of attention.hip:  ab_kernel_flags.sh "" "-DATT_TIMING_HALF_LDS_BWD" ...
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $(extra_flags attention.hip) $flags -c $C/attention.hip -o $C/attention.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
  echo "== flags: '$flags'"
  python scripts/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids | tail -7
done
