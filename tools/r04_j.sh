#!/bin/bash
# round 4, tenth batch: the whole GPU suite on the final library (E12 at five waves per SIMD), c5 with either attention kernel for the question batch
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_j
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
grep -q "pytest rc 0" $O/pytest_gpu.log || exit 1
for v in default mfma default mfma; do
  if [ $v = mfma ]; then export RDX_ENC_MFMA_MIN=1024; else unset RDX_ENC_MFMA_MIN; fi
  timeout -k 10 400 python3 bench.py --workload c5 --no-cpu > $O/c5_$v.json 2> $O/c5_$v.err || echo "c5 $v failed"
  python3 - <<PY
import json
d = json.load(open("$O/c5_$v.json"))
print("c5 $v:", d["value"], "q/s", d["ms_per_step"], "ms; encode avg", d["encode"]["avg_ms"], "serial", d["encode"]["serial_leg"])
PY
done
unset RDX_ENC_MFMA_MIN
timeout -k 10 300 python3 bench.py --workload c2 --no-cpu 2> /dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('c2', d['ms_per_step'], d['roofline'].get('bytes_source'))"
