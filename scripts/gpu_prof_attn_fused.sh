#!/bin/bash
# per-kernel durations of the two forms of the attention backward at the cfg-3 encoder shape (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_attn -- python3 $GRAFT_REPO_ROOT/scripts/check_attn_fused.py --time-only > $GRAFT_REPO_ROOT/gpurun_out/prof_attn.log 2>&1
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/prof_attn.log
find gpurun_out/prof_attn -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/attn_fused_kernel_stats.csv
cut -c1-150 gpurun_out/attn_fused_kernel_stats.csv | head -12
rm -rf gpurun_out/prof_attn
