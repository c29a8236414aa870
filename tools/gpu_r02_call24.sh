#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 20 python -m pytest tests/test_gpu_ipc.py -m gpu -q --timeout 15 2>&1 | tail -3
