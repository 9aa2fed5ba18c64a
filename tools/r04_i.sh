#!/bin/bash
# round 4, ninth batch: E12 with less vector work per score (raw-score maximum, fma exponent, masking on the last tile only, lazy rescale)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_i
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest_enc.log 2>&1; echo "pytest enc rc $?" | tee -a $O/pytest_enc.log
tail -3 $O/pytest_enc.log
grep -q "pytest enc rc 0" $O/pytest_enc.log || exit 1
timeout -k 10 300 python3 tools/attn_bench.py 2>/dev/null | grep -v amdgpu | tee $O/attn_bench.txt
timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee $O/ingest.txt
