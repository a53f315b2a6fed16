for b in 32 64 128 256; do for r in 4 8 16; do GFX_TRACE_BATCH=$b GFX_TRACE_REFILL=$r timeout 300 python bench.py --steps 20 --warmup 5 --mse-ref-spp 0 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_frame']; print('batch $b refill $r', d['ms_per_step'], k['trace_any'], k['trace_closest'])"; done; done
