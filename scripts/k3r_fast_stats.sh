#!/bin/bash
# K3r fast body: per-query statistics of the published floor and the wave iterations it lets skip (comparison flavour: ORAMA_K3R_DBG=16 + ORAMA_K3R_STATS=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ORAMA_COMPARISON_KERNELS=1 ORAMA_K3R_FAST=1 ORAMA_BM25_DENSE_ACC=1 ORAMA_K3R_DBG=16 ORAMA_K3R_STATS=1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc --details-file /tmp/k3r_stats.json 2>&1 >/dev/null | grep "^\[k3r\]" | head -30
