#!/bin/bash
# round 4, sixth batch: MFMA attention for large question batches + in-place GELU (E13): encoder tests, GEMM+GELU table, c5, ingest
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_f
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
timeout -k 10 300 python3 tools/gemm_layouts.py 20480 2>/dev/null | grep -v amdgpu | tee $O/gemm_layouts.txt
timeout -k 10 600 python3 bench.py --workload c5 --no-others --steps 30 --warmup 8 > $O/c5.json 2> $O/c5.err || echo "c5 failed"
python3 - <<PY
import json
d = json.load(open("$O/c5.json"))
print("c5", d["value"], "q/s", d["ms_per_step"], "ms", d.get("encode"))
PY
RDX_ENC_GELU=torch timeout -k 10 600 python3 bench.py --workload c5 --no-others --steps 30 --warmup 8 > $O/c5_torch_gelu.json 2> $O/c5_torch_gelu.err || echo "c5 torch gelu failed"
python3 - <<PY
import json
d = json.load(open("$O/c5_torch_gelu.json"))
print("c5 torch gelu", d["value"], "q/s", d["ms_per_step"], "ms")
PY
timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee $O/ingest.txt
