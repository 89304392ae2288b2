"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes: the TCC
block has 4 counter slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").

On the GPU box (one command per pass, program directly after `--`, no trace domains next to --pmc):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 \
        --no-cpu-baseline --no-roofline --no-host-leg
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py ...   (same)
then here:
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r2_pmc_traffic.json

Units and the gfx950 correction (MI355X_MICROARCH.md "HBM"): both counters are in KiB; FETCH_SIZE reports exactly half of
the bytes of wide (16 B per lane) coalesced reads -- every read of the kernels below is a 16-byte load or an LDS-DMA of 16
bytes per lane -- so  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Infinity-Cache hits are counted, not excluded.
Per kernel symbol the per-launch mean over all its dispatches in the run is written, next to the raw counter means, and
the same split by launch geometry (`by_grid`: the ViT self-attention and the decoder's cross-attention launches of one symbol move
different byte counts -- a mean over the mix is not comparable with either shape's algorithmic bytes).  `csrc_sha1` records the
kernel sources the passes were taken from: bench.py drops `roofline.traffic` when the source of the dominant kernel has changed."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    """'void (anonymous namespace)::attn_fwd_kernel<false>((anonymous namespace)::AttnArgs)' -> 'attn_fwd_kernel<false>'"""
    n = re.sub(r'\(anonymous namespace\)::', '', name)
    n = re.sub(r'^void\s+', '', n)
    depth, out = 0, ''
    for ch in n:            # cut the argument list: the first '(' at template depth 0
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out += ch
    out = out.strip()
    m = re.match(r'^(attn_\w+)<(\w+)(?:,[^>]*)?>$', out)      # attention kernels: <causal, dropout, prescaled-q> -> <causal> (bench.py's names)
    return f'{m.group(1)}<{m.group(2)}>' if m else out


# kernels whose grid does not tell their launch shapes apart, and the kernel launched right before them in the same crl_* call whose
# grid does: the dK/dV pass has one workgroup per 128 KEYS (6189 encoder keys for the ViT self-attention and for the decoder's
# cross-attention alike); the dQ pass that precedes it has one per 128 QUERIES (6189 vs 1023)
SHAPE_PARTNER = {'attn_bwd_dkdv_kernel<false>': 'attn_bwd_dq_kernel<false>', 'attn_bwd_dkdv_kernel<true>': 'attn_bwd_dq_kernel<true>'}


def read_pass(d, counter):
    """{(kernel, shape key): (mean counter value, launches)}; shape key None = all launches of the symbol, otherwise the grid size in
    threads of the launch (of its SHAPE_PARTNER for the kernels listed there)"""
    sums, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    assert files, f'no counter_collection.csv under {d}'
    for f in files:
        with open(f, newline='') as fh:
            rows = [r for r in csv.DictReader(fh) if r['Counter_Name'] == counter]
        rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0) or 0))
        last_grid = {}
        for row in rows:
            k = short(row['Kernel_Name'])
            grid = int(row.get('Grid_Size', 0) or 0)
            last_grid[k] = grid
            shape = last_grid.get(SHAPE_PARTNER[k], grid) if k in SHAPE_PARTNER else grid
            for key in ((k, None), (k, shape)):
                sums[key] += float(row['Counter_Value'])
                cnt[key] += 1
    return {k: (sums[k] / cnt[k], cnt[k]) for k in sums}


def csrc_sha1():
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pixparse_amd', 'csrc')
    return {os.path.basename(f): hashlib.sha1(open(f, 'rb').read()).hexdigest() for f in sorted(glob.glob(os.path.join(root, '*.hip')) + glob.glob(os.path.join(root, '*.h')))}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = read_pass(fetch_dir, 'FETCH_SIZE'), read_pass(write_dir, 'WRITE_SIZE')
    kernels = {}
    for key in sorted(set(fe) & set(wr), key=lambda kg: (kg[0], kg[1] or 0)):
        k, grid = key
        f, nf = fe[key]
        w, nw = wr[key]
        rec = {'launches': min(nf, nw), 'FETCH_SIZE_KiB': round(f, 1), 'WRITE_SIZE_KiB': round(w, 1),
               'hbm_bytes_per_launch': int((2.0 * f + w) * 1024)}
        if grid is None:
            kernels.setdefault(k, {}).update(rec)
        else:
            kernels.setdefault(k, {}).setdefault('by_grid', {})[str(grid)] = rec
    res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) over `bench.py --steps 1 --warmup 1`, cfg-3; '
                     'hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request); '
                     + os.path.basename(out),
           'csrc_sha1': csrc_sha1(), 'kernels': kernels}
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:25]:
        print(f"{k[:70]:70s} n={v['launches']:4d}  {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == '__main__':
    main()
