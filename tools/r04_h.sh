#!/bin/bash
# round 4, eighth batch: the in-place GELU (E13) — its test, FFN1 timing, c5 with it and with the framework's GELU, ingest
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_h
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest_enc.log 2>&1; echo "pytest enc rc $?" | tee -a $O/pytest_enc.log
tail -4 $O/pytest_enc.log
grep -q "pytest enc rc 0" $O/pytest_enc.log || exit 1
timeout -k 10 300 python3 tools/gemm_layouts.py 20480 2>/dev/null | grep -v amdgpu | tee $O/gemm_layouts.txt
for g in inplace torch inplace torch; do
  RDX_ENC_GELU=$g timeout -k 10 600 python3 bench.py --workload c5 --no-others --no-cpu --steps 30 --warmup 8 > $O/c5_$g.json 2> $O/c5_$g.err || echo "c5 $g failed"
  python3 - <<PY
import json
d = json.load(open("$O/c5_$g.json"))
print("c5 gelu $g:", d["value"], "q/s", d["ms_per_step"], "ms; encode avg", d["encode"]["avg_ms"], "serial", d["encode"]["serial_leg"], "host", d["encode"]["host_ms_per_step"])
PY
done
timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee $O/ingest.txt
RDX_ENC_GELU=torch timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee $O/ingest_torch_gelu.txt
