#!/bin/bash
# PMC passes over the attention micro-benchmark (scripts/bench_kernels.py attn: B 8, H 16, N 6189, d 64, random data): effective clock,
# matrix-pipe busy cycles, VALU issue, co-execution, LDS activity / bank conflicts -- one rocprofv3 run per counter group (no trace domains).
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
export TAG
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_attn_$i -- python3 $R/scripts/bench_kernels.py attn > $R/gpurun_out/pmc_attn_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_attn_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '').split('(')[0]
        if not n.startswith('attn'): continue
        if int(r['Grid_Size']) != 6272 * 256 and not n.startswith(('attn_dq_reduce', 'attn_bwd_spx')): continue     # ViT-shape launches: 128-row tiles x 128 heads (the single pass: chains of 256-key blocks x 128 heads)
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
import os
with open('gpurun_out/%s_pmc_attention.txt' % os.environ.get('TAG', 'r3'), 'w') as out:
    for n, c in sorted(agg.items()):
        line = f'{n}: ' + '  '.join(f'{k}={sum(v) / len(v):.4g} (n={len(v)})' for k, v in sorted(c.items()))
        print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/pmc_attn_*/
