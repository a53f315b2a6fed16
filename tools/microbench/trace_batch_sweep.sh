for b in 16 64 256 1024 4096; do echo "batch $b"; GFX_TRACE_BATCH=$b GFX_TEMPORAL_HINTS=0 python tools/trace_tail.py 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:(v.get('natural_ms')) for k,v in d.items()})"; done
