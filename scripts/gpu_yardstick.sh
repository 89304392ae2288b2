#!/bin/bash
# vendor GEMM / SDPA against this library on one box (table), then the clocks of the same kernels (GRBM_GUI_ACTIVE pass)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
if [ "$1" != pmconly ]; then python3 scripts/yardstick.py > gpurun_out/r5_yardstick.txt 2>&1; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/yard_pmc -- python3 $R/scripts/yardstick.py pmc > $R/gpurun_out/r5_yardstick_pmc.log 2>&1
cd $R
python3 - <<'PY' >> gpurun_out/r5_yardstick.txt
import csv, glob, collections
agg = collections.defaultdict(list)
meta = {}
for f in glob.glob('gpurun_out/yard_pmc/**/*counter_collection.csv', recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd:
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        ns = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        if ns < 50000: continue
        key = (r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:110], r["Grid_Size"])
        agg[key].append((float(r['Counter_Value']) / 8.0 / ns, ns))
        meta[key] = 'wg %s lds %s vgpr %s agpr %s sgpr %s' % (r.get('Workgroup_Size'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'))
print('\n== effective clock (GRBM_GUI_ACTIVE / 8 / duration; profiled pass, 6 launches per row: reads high on short dispatches) ==')
for key, v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    n, grid = key
    print(f'{n}\n      grid {grid:>9s} {meta[key]}  n={len(v):3d} avg {sum(x[1] for x in v)/len(v)/1000:9.1f} us  clock {sum(x[0] for x in v)/len(v):.3f} GHz')
PY
head -3 $(find gpurun_out/yard_pmc -name '*counter_collection.csv' | head -1) > gpurun_out/r5_yardstick_pmc_head.txt; rm -rf gpurun_out/yard_pmc
