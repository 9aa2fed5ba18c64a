#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_check3
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for rnd in 1 2; do for ff in 1 0; do for w in c1 c2 c3; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu --steps 300 --warmup 30 --set fuse_finish=$ff > $O/${w}_ff${ff}_$rnd.json 2> /dev/null || echo "bench $w failed"
done; done; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"] and d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
