#!/bin/bash
# per-kernel times of the single-pass attention backward at several chain lengths (rocprofv3 kernel trace, one run per chain length)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chain_$c -- python3 $R/scripts/ab_chain.py $c > $R/gpurun_out/chain_$c.log 2>&1
  echo "== chain $c"
  python3 - $R/gpurun_out/chain_$c <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'attn_bwd_spx' in n or 'dq_reduce' in n:
            agg[(n.split('(')[0][-40:], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v)
    print(f'  {k[0]:40s} grid {k[1]:>9s} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}')
PY
  rm -rf $R/gpurun_out/chain_$c
done
