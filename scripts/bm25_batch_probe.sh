python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['c4']; b=c['bm25_only']
print('bm25 batch', round(b['value']), 'best', round(b['runs']['best']), 'single', round(b['single_query_calls']['value']), 'dev us', b['roofline']['device_us_by_kernel'], '| hybrid', round(c['value'],2), 'p50', round(c['latency_ms_p50'],4))"
