#!/bin/bash
# times every variant library under pixparse_amd/csrc/variants (scripts/ab_f4w.sh) on the ViT forward shape, product library first and last
ulimit -c 0
mkdir -p gpurun_out
out=gpurun_out/f4w_variants.txt
: > $out
run() { echo "== $1" >> $out; PIXPARSE_AMD_LIB=$2 timeout 120 python scripts/bench_attn_fwd.py $BENCH_ARGS 2>&1 | grep -v amdgpu.ids >> $out; }
run product pixparse_amd/csrc/libcruller_hip.so
for v in pixparse_amd/csrc/variants/*.so; do run $(basename $v .so) $v; done
run product pixparse_amd/csrc/libcruller_hip.so
cat $out
