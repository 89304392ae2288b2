#!/bin/bash
# same-box A/B of compile-time knobs of ONE csrc file on the kernel micro-benchmarks:
#   ab_kernels.sh gemm4w.hip gemm4w "" "-DG4_CU_STAGGER=0" "-DG4_CU_STAGGER=1"
# (file, scripts/bench_kernels.py mode, grep pattern, flag sets...)
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
F=$1; MODE=$2; PAT=$3; shift 3
OBJ=$C/${F%.*}.o
EXTRA="$(extra_flags $F)"
OBJS=$(ls $C/*.o | tr '\n' ' ')
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA $flags -c $C/$F -o $OBJ || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $OBJS || exit 1
  echo "== $flags"
  python scripts/bench_kernels.py $MODE 2>&1 | grep -E "$PAT"
done
