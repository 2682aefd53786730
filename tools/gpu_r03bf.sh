#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_host_cli.py -x -q 2>&1 | tail -5
