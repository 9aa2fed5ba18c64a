#!/bin/bash
# round 4, seventh batch: the one-wave-per-SIMD scan (wave_layout = 1): parity first, then c4 beside the shipped kernel; then batch f's legs
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_g
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "one_wave_per_simd or fused_epilogue" > $O/pytest_w4.log 2>&1; echo "pytest w4 rc $?" | tee -a $O/pytest_w4.log
tail -5 $O/pytest_w4.log
if grep -q "pytest w4 rc 0" $O/pytest_w4.log; then
  for rep in 1 2; do for wl in 0 1; do
    timeout -k 10 500 python3 bench.py --workload c4 --no-others --no-cpu --steps 40 --warmup 10 --set wave_layout=$wl > $O/c4_wl${wl}_$rep.json 2> $O/c4_wl${wl}_$rep.err || echo "c4 wl $wl failed"
    python3 - <<PY
import json
d = json.load(open("$O/c4_wl${wl}_$rep.json"))
print("c4 wave_layout $wl rep $rep:", d["value"], "q/s", d["ms_per_step"], "ms", d["roofline"])
PY
  done; done
fi
timeout -k 10 900 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest_enc.log 2>&1; echo "pytest enc rc $?" | tee -a $O/pytest_enc.log
tail -4 $O/pytest_enc.log
grep -q "pytest enc rc 0" $O/pytest_enc.log || exit 1
timeout -k 10 300 python3 tools/gemm_layouts.py 20480 2>/dev/null | grep -v amdgpu | tee $O/gemm_layouts.txt
timeout -k 10 600 python3 bench.py --workload c5 --no-others --steps 30 --warmup 8 > $O/c5.json 2> $O/c5.err || echo "c5 failed"
python3 - <<PY
import json
d = json.load(open("$O/c5.json"))
print("c5", d["value"], "q/s", d["ms_per_step"], "ms", d.get("encode"))
PY
