#!/bin/bash
# the query split of the attention backward's remainder chains and the persistent (ticket-pulling) launch in the cfg-3 step, alternating on one box
ulimit -c 0
mkdir -p gpurun_out
out=gpurun_out/r5_attn_bwd_qsplit_step_ab.txt
: > $out
for rep in 1 2 3; do
for kv in "PIXPARSE_AMD_ATTN_BWD_QSPLIT=0 PIXPARSE_AMD_ATTN_BWD_PERSIST=0" "PIXPARSE_AMD_ATTN_BWD_QSPLIT=-1 PIXPARSE_AMD_ATTN_BWD_PERSIST=0" "PIXPARSE_AMD_ATTN_BWD_QSPLIT=-1 PIXPARSE_AMD_ATTN_BWD_PERSIST=1"; do
  echo "== $kv: $(env $kv python bench.py --no-cpu-baseline --no-roofline --no-host-leg --no-peak --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step", "loss", d["loss"])')" >> $out
done; done
cat $out
