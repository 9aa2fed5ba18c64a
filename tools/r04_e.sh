#!/bin/bash
# round 4, fifth batch: GPU suite on the refactored refine + adaptive sample + pipelined ingest; ingest; the two corpus shapes with defaults
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_e
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
RDX_FUZZ_SEED=4201 RDX_FUZZ_CASES=250 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz_4201.log 2>&1; echo "fuzz 4201 rc $?" | tee -a $O/pytest.log
timeout -k 10 300 python3 tools/ingest_host_profile.py 3000 2>/dev/null | grep -v amdgpu | tee $O/ingest_host_profile.txt
timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee $O/ingest.txt
for w in c3 c4; do for shape in embed iid; do
  timeout -k 10 600 python3 bench.py --workload $w --corpus-shape $shape --no-others --steps 40 --warmup 10 --profile-all > $O/${w}_${shape}.json 2> $O/${w}_${shape}.err || echo "bench $w $shape failed"
  python3 - <<PY
import json
d = json.load(open("$O/${w}_${shape}.json"))
print("$w $shape", d["value"], "q/s", d["ms_per_step"], "ms", d["path_stats"], (d["recall_at_10"] or {}).get("ids_bit_exact"), (d["recall_at_10"] or {}).get("full_corpus_exact_scan_equals_mfma_path"))
PY
done; done
timeout -k 10 300 python3 tools/gemm_layouts.py 20480 2>/dev/null | grep -v amdgpu | tee $O/gemm_layouts.txt
timeout -k 10 300 python3 tools/attn_bench.py 2>/dev/null | grep -v amdgpu | tee $O/attn_bench.txt
