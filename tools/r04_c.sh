#!/bin/bash
# round 4, third batch: the whole GPU suite on the new refine / sample code, fuzz hunts, the embedding-like corpus again
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
for seed in 4101 4102; do
  RDX_FUZZ_SEED=$seed RDX_FUZZ_CASES=200 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz_$seed.log 2>&1; echo "fuzz $seed rc $?" | tee -a $O/pytest.log; tail -2 $O/fuzz_$seed.log
done
for w in c3 c4; do for shape in embed iid; do
  timeout -k 10 600 python3 bench.py --workload $w --corpus-shape $shape --no-others --steps 30 --warmup 10 > $O/${w}_${shape}.json 2> $O/${w}_${shape}.err || echo "bench $w $shape failed"
  python3 - <<PY
import json
d = json.load(open("$O/${w}_${shape}.json"))
print("$w $shape", d["value"], "q/s", d["ms_per_step"], "ms", "frac", d["roofline"]["frac"] if d["roofline"] else None, d["path_stats"], d["recall_at_10"])
PY
done; done
timeout -k 10 600 python3 bench.py --workload c4 --corpus-shape embed --no-others --steps 30 --warmup 10 --set spread_boot=0 > $O/c4_embed_tile_sample.json 2> /dev/null || echo "failed"
python3 - <<PY
import json
d = json.load(open("$O/c4_embed_tile_sample.json"))
print("c4 embed spread_boot=0", d["value"], "q/s", d["ms_per_step"], "ms", d["path_stats"])
PY
