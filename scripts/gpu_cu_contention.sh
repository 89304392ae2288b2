#!/bin/bash
# Same-box A/B of the cfg-3 train step with CUs taken away by a sleeping side-stream kernel (crl_debug_occupy_cus): the single-GPU
# stand-in for the CUs RCCL's all-reduce kernels hold in a data-parallel run.
#   static   = the persistent GEMMs walk fixed tile lists (workgroup b: b, b + grid, ...)
#   dynamic  = resident workgroups pull tiles from ticket counters (default)
#   reserved = dynamic + crl_gemm_set_reserved_cus(n): launch on 256 - n CUs, wave-quantisation split re-planned for that width
# Output: one line per run in gpurun_out/cu_contention.txt (copy to profiles/).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/cu_contention.txt
: > $out
run() {  # label, bench args...
  local label=$1; shift
  local line
  line=$(python bench.py --no-cpu-baseline --no-roofline --no-host-leg --no-peak --steps ${STEPS:-6} --warmup 2 "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "docs/s loss", d["loss"])')" | tee -a $out
}
run "occupied=0   schedule=static           " --gemm-schedule static
run "occupied=0   schedule=dynamic          " --gemm-schedule dynamic
for n in ${CUS:-16 32 64}; do
  run "occupied=$n  schedule=static           " --occupy-cus $n --gemm-schedule static
  run "occupied=$n  schedule=dynamic          " --occupy-cus $n --gemm-schedule dynamic
  run "occupied=$n  schedule=dynamic reserved=$n" --occupy-cus $n --gemm-schedule dynamic --reserved-cus $n
done
run "occupied=0   schedule=dynamic (repeat) " --gemm-schedule dynamic
