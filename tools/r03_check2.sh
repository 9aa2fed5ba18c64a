#!/bin/bash
# parity first, then the lines the change is about
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_check2
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for w in c3 c2; do timeout -k 10 300 python3 bench.py --workload $w --no-cpu --steps 200 --warmup 20 --profile-all > $O/$w.json 2> $O/$w.err || echo "bench $w failed"; done
timeout -k 10 300 python3 bench.py --workload c4 --no-cpu --rows 1250000 --steps 30 --warmup 5 --profile-all > $O/c4_eighth.json 2> /dev/null; echo "eighth rc=$?"
timeout -k 10 300 python3 bench.py --workload c4 --no-cpu --steps 10 --warmup 3 > $O/c4.json 2> /dev/null; echo "c4 rc=$?"
timeout -k 10 300 python3 bench.py --workload c4 --no-cpu --steps 10 --warmup 3 --set spec_tau=0 > $O/c4_nospec.json 2> /dev/null; echo "c4 nospec rc=$?"
timeout -k 10 300 python3 bench.py --workload c3 --no-cpu --steps 200 --warmup 20 --profile-all --set spec_tau=0 > $O/c3_nospec.json 2> /dev/null
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f)); ps = d["path_stats"]; print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"] and d["roofline"]["frac"], "emit/q", ps["emitted_per_query"], "resc/q", ps["rescored_per_query"], "rank", ps["tau_rank"], "retried", ps["retried_queries"], ps["ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
