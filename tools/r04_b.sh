#!/bin/bash
# round 4, second batch: the trimmed single-question path, the ingest host profile, the new bench legs, the embedding-like corpus
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_embedding_provider.py tests/test_gpu_cabi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout -k 10 200 python3 tools/enc_small_bench.py 20 2>/dev/null | tee $O/enc_small.txt
timeout -k 10 300 python3 tools/enc_single.py 2>/dev/null | grep -v amdgpu | tee $O/enc_single.txt
timeout -k 10 300 python3 tools/collection_latency.py --online 2>/dev/null | grep -v amdgpu | tee $O/collection_latency.txt
timeout -k 10 400 python3 tools/ingest_host_profile.py 1600 2>/dev/null | grep -v amdgpu > $O/ingest_host_profile.txt; head -60 $O/ingest_host_profile.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/c4_with_others.json 2> $O/c4.err || echo "bench failed"
python3 - <<PY
import json
d = json.load(open("$O/c4_with_others.json"))
print("c4", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d["other_configs"], indent=1)[:3000])
PY
for w in c3 c4; do for shape in iid embed; do
  timeout -k 10 600 python3 bench.py --workload $w --corpus-shape $shape --no-others > $O/${w}_${shape}.json 2> $O/${w}_${shape}.err || echo "bench $w $shape failed"
  python3 - <<PY
import json
d = json.load(open("$O/${w}_${shape}.json"))
print("$w $shape", d["value"], "q/s", d["ms_per_step"], "ms", "frac", d["roofline"]["frac"] if d["roofline"] else None, d["path_stats"], d["recall_at_10"])
PY
done; done
