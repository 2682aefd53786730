#!/bin/bash
# the whole GPU suite and the smoke entry at the round's last tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r03cl_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
