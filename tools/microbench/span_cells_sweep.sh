for c in 1 2 4 8 16 32; do
  for f in "" "--plain"; do
    GFX_SPAN_CELLS_PER_REC=$c timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cells/rec $c $f', d['ms_per_step'], d['kernels_ms_per_frame']['initial_candidates'])"
  done
done
