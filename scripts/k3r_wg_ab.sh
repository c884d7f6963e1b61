#!/bin/bash
# K3r: scoring workgroups of 512 threads (ranges of <= 4 096 postings, target 3 584) against 256 (2 048 / 1 792): the library in the
# tree is whatever ORAMA_K3R_WG built; this prints the C4 batch figures + the full-text GPU tests on it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python scripts/k3r_chunk_probe.py 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc --details-file /tmp/k3r_wg.json 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['configs']['bm25_batch']; print('bm25 batch', json.dumps(b)[:400]); print('c4', json.dumps(d['configs']['c4'])[:300])"
python - <<'PY'
import json
d = json.load(open('/tmp/k3r_wg.json'))
b = d['configs']['c4']['bm25_batch'] if 'bm25_batch' in d['configs']['c4'] else d['configs'].get('bm25_batch')
print(json.dumps(b)[:900])
PY
[ "$1" = tests ] && timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "bm25 or token or fulltext or reference or hybrid or post or filter or facet or shard" 2>&1 | tail -4
