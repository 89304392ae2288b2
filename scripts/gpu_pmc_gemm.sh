#!/bin/bash
# PMC passes over the 8192^3 GEMM: the 4-wave kernel (overlapped form, what the auto policy picks) against the 8-wave kernel -- effective clock,
# matrix-pipe busy cycles, LDS activity, waits.  One rocprofv3 run per counter group (no trace domains).  Output: gpurun_out/<tag>_pmc_gemm.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-r5}
export TAG
mkdir -p $R/gpurun_out
cat > /tmp/pmc_gemm_case.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
n = 8192
x = torch.randn(n, n, device=dev).to(BF16); w = torch.randn(n, n, device=dev).to(BF16); out = torch.empty(n, n, dtype=BF16, device=dev)
hip.call('crl_gemm_set_policy', 2)
for big in (0, 1):
    hip.call('crl_gemm_set_big_kernel', big)
    for _ in range(40): ops.linear_fwd(x, w, None, out)
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_gemm_$i -- python3 /tmp/pmc_gemm_case.py > $R/gpurun_out/pmc_gemm_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, re, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_gemm_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '').split('(')[0]
        if not n.startswith('gemm'): continue
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open('gpurun_out/%s_pmc_gemm.txt' % os.environ.get('TAG', 'r5'), 'w') as out:
    for n, c in sorted(agg.items()):
        d = sum(dur[n]) / len(dur[n])
        line = f'{n}: avg {d / 1e3:.1f} us (profiled)  ' + '  '.join(f'{k}={sum(v) / len(v):.4g}' for k, v in sorted(c.items()))
        if 'GRBM_GUI_ACTIVE' in c:
            line += f'  | clock {sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8 / d:.3f} GHz'
        print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/pmc_gemm_*/
