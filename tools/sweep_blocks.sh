cd $GRAFT_REPO_ROOT
for b in 1 2 3 4 6; do
  echo "== blocks/CU $b"
  GFX_TRACE_BLOCKS_PER_CU=$b python bench.py --steps 15 --warmup 3 --mse-ref-spp 0 --cpu-sample 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_frame'].items() if 'trace' in k}, d['roofline']['frac'], d['roofline']['scheduling'])"
done
