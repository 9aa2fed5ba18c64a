#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_check5
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_k or clustered or fast_path or baseline_config2" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for rnd in 1 2; do for sb in 1 0; do for w in c2; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu --steps 300 --warmup 30 --set split_boot=$sb > $O/${w}_sb${sb}_$rnd.json 2> /dev/null || echo "bench $w failed"
done; done; done
timeout -k 10 300 python3 bench.py --workload c2 --no-cpu --steps 300 --warmup 30 --profile-all > $O/c2_profall.json 2> /dev/null
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f)); ps = d["path_stats"]; print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"] and d["roofline"]["frac"], "emit/q", ps["emitted_per_query"], "rank", ps["tau_rank"], "sample", ps["sample_rows"], ps["ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
