#!/bin/bash
# fused encoder kernels: GPU tests, then A/B of the c5 step (fused / torch-ops packed / module forward) and a kernel breakdown
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_enc
mkdir -p $O && cd $R
timeout -k 10 500 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 300 python3 tools/enc_ab.py > $O/enc_ab.txt 2>&1 || { tail -20 $O/enc_ab.txt; exit 1; }
cat $O/enc_ab.txt
