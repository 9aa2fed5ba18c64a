#!/bin/bash
# round 4, fourth batch: ingest after the host-side numpy fix (+ where a batch's wall time goes), config 5 with canonical-shape graphs for
# large batches, k_refine's phases at c3 / c2 shapes
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/ingest_host_profile.py 3000 2>/dev/null | grep -v amdgpu | tee $O/ingest_host_profile.txt
timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee $O/ingest.txt
RDX_ENC_BATCH=128 timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee -a $O/ingest.txt
for lg in 1 0; do
  RDX_ENC_LARGE_GRAPHS=$lg timeout -k 10 500 python3 bench.py --workload c5 --no-cpu > $O/c5_large_graphs_$lg.json 2> $O/c5_$lg.err || echo "c5 failed"
  python3 - <<PY
import json
d = json.load(open("$O/c5_large_graphs_$lg.json"))
print("c5 large_graphs=$lg", d["value"], "q/s", d["ms_per_step"], "ms/step; encode", d["encode"]["avg_ms"], d["encode"]["host_ms_per_step"], d["encode"]["serial_leg"])
PY
done
python3 tools/ab_lib.py build rstamps "-DRDX_REFINE_STAMPS" > $O/build_rstamps.log 2>&1 && {
  timeout -k 10 300 python3 tools/refine_stamps.py 1000000 256 100 2>/dev/null | grep -v amdgpu | tee $O/refine_stamps.txt
  timeout -k 10 300 python3 tools/refine_stamps.py 100000 64 10 2>/dev/null | grep -v amdgpu | tee -a $O/refine_stamps.txt
  timeout -k 10 300 python3 tools/refine_stamps.py 1250000 1024 10 2>/dev/null | grep -v amdgpu | tee -a $O/refine_stamps.txt
}
for sd in 64 32 16; do
  timeout -k 10 600 python3 bench.py --workload c4 --corpus-shape embed --no-others --no-cpu --steps 20 --warmup 8 --profile-all --set sample_div=$sd > $O/c4_embed_div$sd.json 2> /dev/null || echo failed
  python3 - <<PY
import json
d = json.load(open("$O/c4_embed_div$sd.json"))
print("c4 embed sample_div=$sd", d["value"], "q/s", d["ms_per_step"], "ms", d["path_stats"])
PY
done
timeout -k 10 600 python3 bench.py --workload c4 --no-others --no-cpu --steps 20 --warmup 8 --profile-all --set sample_div=32 > $O/c4_iid_div32.json 2> /dev/null || echo failed
python3 - <<PY
import json
d = json.load(open("$O/c4_iid_div32.json"))
print("c4 iid sample_div=32", d["value"], "q/s", d["ms_per_step"], "ms", d["path_stats"])
PY
