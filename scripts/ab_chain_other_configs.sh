for spec in "cruller_base_960x640 8" "cruller_large_6layers 2"; do
  set -- $spec
  for rep in 1 2; do
  for c in 1 0; do
    PIXPARSE_AMD_ATTN_BWD_CHAIN=$c python bench.py --model $1 --batch $2 --no-cpu-baseline --no-roofline --no-host-leg --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 batch $2 chain=$c:', d['value'], 'docs/s', d['ms_per_step'], 'ms/step', 'loss', d['loss'])"
  done; done
done
