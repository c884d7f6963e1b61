#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vector_f16_gpu.py -m gpu -x -q -p no:cacheprovider -k "super_chunks" 2>&1 | tail -15 | cut -c1-220
