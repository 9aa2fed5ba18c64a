#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c5b
mkdir -p $O; cd $R
for rnd in 1 2; do for v in packed module; do
  if [ $v = module ]; then F=--enc-module-forward; else F=; fi
  timeout -k 10 400 python3 bench.py --workload c5 --no-cpu $F > $O/c5_${v}_$rnd.json 2> /dev/null || echo fail $v
done; done
python3 - <<PY
import json
for rnd in (1,2):
    for v in ("packed","module"):
        d=json.load(open("$O/c5_%s_%d.json"%(v,rnd))); e=d["encode"]
        print(rnd, v, d["value"], d["ms_per_step"], "enc", e["avg_ms"], "serial", e["serial_leg"], e["length_buckets"]["buckets"])
PY
