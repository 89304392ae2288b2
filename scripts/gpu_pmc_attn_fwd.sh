#!/bin/bash
# PMC passes over the forward attention at the ViT shape (B 8, H 16, N 6189, d 64, random data): the compiler-scheduled kernel against the hand-placed
# stream at one / two workgroups per CU -- effective clock, matrix-pipe busy cycles, VALU / LDS instructions, LDS activity, waits, HBM bytes.
# One rocprofv3 run per counter group (no trace domains).  Output: gpurun_out/<tag>_pmc_attn_fwd.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-r5}
export TAG
mkdir -p $R/gpurun_out
cat > /tmp/pmc_attn_fwd_case.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pixparse_amd import hip, ops
dev = torch.device('cuda:0')
B, H, N = 8, 16, 6189
D = H * 64
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(B, N, 3 * D, generator=g, device=dev).to(torch.bfloat16)
qp = (qkv[:, :, :D].float() * 0.125 * ops.LOG2E).to(torch.bfloat16)
o = torch.empty(B, N, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, H, N, device=dev)
for mode in (1, 3, 0):
    hip.call('crl_attn_fwd_set_mode', mode)
    for _ in range(30): ops.attn_fwd(qp, qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, lse, H, 0.125, False, q_prescaled=True)
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_af_$i -- python3 /tmp/pmc_attn_fwd_case.py > $R/gpurun_out/pmc_af_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, re, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_af_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '').split('(')[0]
        if not n.startswith('attn_fwd'): continue
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open('gpurun_out/%s_pmc_attn_fwd.txt' % os.environ.get('TAG', 'r5'), 'w') as out:
    for n, c in sorted(agg.items()):
        d = sum(dur[n]) / len(dur[n])
        line = f'{n}: avg {d / 1e3:.1f} us (profiled)  ' + '  '.join(f'{k}={sum(v) / len(v):.4g}' for k, v in sorted(c.items()))
        if 'GRBM_GUI_ACTIVE' in c:
            line += f'  | clock {sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8 / d:.3f} GHz'
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:      # KiB; gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md HBM)
            fs, ws = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']), sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
            line += f'  | HBM bytes per launch {(2 * fs + ws) * 1024 / 1e6:.1f} MB (algorithmic q, k, v, o once: {4 * 8 * 6189 * 1024 * 2 / 1e6:.1f} MB)'
        print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/pmc_af_*/
