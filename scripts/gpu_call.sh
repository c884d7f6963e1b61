#!/bin/bash
# one GPU call: the whole parity suite
O=gpurun_out/r02full
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
