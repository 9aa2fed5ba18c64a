#!/bin/bash
# developer: the one-wave-per-SIMD scan against the oracle on a few shapes (tools/w4_debug.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for shape in "70000 1024 600" "70000 128 600" "70000 1024 1024"; do
  echo "=== $shape"
  timeout -k 10 200 python3 tools/w4_debug.py $shape 10 2>&1 | grep -v amdgpu | grep "wave_layout 1\|missing rows\|query block" | cut -c1-150
done
