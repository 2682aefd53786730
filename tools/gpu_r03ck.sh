#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "sort_variants or fullsize_matches_oracle" 2>&1 | tail -3
