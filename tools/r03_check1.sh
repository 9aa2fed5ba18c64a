#!/bin/bash
# round 3, first GPU contact: the GPU suite on the new ABI, then the N>1 path through the rehearsal (gloo, both ranks on cuda:0)
# and through a one-rank RCCL group, then the baseline lines to compare later kernel work against.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_check1
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
RDX_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --rows 600000 --steps 3 --warmup 1 --check-merged > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearsal rc=$?"
timeout -k 10 300 python3 bench.py --workload c4 --no-cpu --rows 1250000 --steps 30 --warmup 5 --force-dist > $O/c4_eighth_rccl1.json 2> $O/c4_eighth_rccl1.err; echo "force-dist rc=$?"
timeout -k 10 300 python3 bench.py --workload c4 --no-cpu --rows 1250000 --steps 30 --warmup 5 > $O/c4_eighth.json 2> /dev/null; echo "eighth rc=$?"
for w in c3 c2 c1; do timeout -k 10 300 python3 bench.py --workload $w --no-cpu --steps 200 --warmup 20 --profile-all > $O/$w.json 2> $O/$w.err || echo "bench $w failed"; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"] and d["roofline"]["frac"], d.get("merged_equals_single_index"), d.get("distributed_check"), d.get("step_breakdown"), d["path_stats"]["ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
