#!/bin/bash
# K3r: postings per document range (option k3r_target, default 1 536) against the BM25 batch rate of bench.py's C4 leg — comparison flavour (env options)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ORAMA_COMPARISON_KERNELS=1
for T in 1536 1024 1280 1792 1920; do
  echo "== ORAMA_K3R_TARGET=$T"
  ORAMA_K3R_TARGET=$T timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc --details-file /tmp/k3r_$T.json 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['configs']['bm25_batch']; print('bm25 batch', b['value'], 'q/s, device us/query', b['us_per_query_device'], 'single calls/s', b['single_calls_per_s'], '| c4', d['configs']['c4']['value'])"
done
