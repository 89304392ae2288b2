#!/bin/bash
# per-kernel durations of the attention backward forms at the cfg-3 shapes (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_attn -- python3 $GRAFT_REPO_ROOT/scripts/check_attn_sp.py --time-only > $GRAFT_REPO_ROOT/gpurun_out/prof_attn_sp.log 2>&1
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/prof_attn_sp.log
find gpurun_out/prof_attn -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/attn_sp_kernel_stats.csv
cut -c1-160 gpurun_out/attn_sp_kernel_stats.csv | head -12
rm -rf gpurun_out/prof_attn
