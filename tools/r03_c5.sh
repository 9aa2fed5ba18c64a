#!/bin/bash
# BASELINE config 5 (query encode in the step): the packed forward with librdx's encoder kernels / on torch operations / the module forward, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c5c
mkdir -p $O; cd $R
for rnd in 1 2; do for v in fused packed module; do
  case $v in module) F=--enc-module-forward;; packed) F=--enc-torch-ops;; *) F=;; esac
  timeout -k 10 400 python3 bench.py --workload c5 --no-cpu $F > $O/c5_${v}_$rnd.json 2> /dev/null || echo fail $v
done; done
python3 - <<PY
import json
for rnd in (1,2):
    for v in ("fused","packed","module"):
        d=json.load(open("$O/c5_%s_%d.json"%(v,rnd))); e=d["encode"]
        print(rnd, v, d["value"], d["ms_per_step"], "enc", e["avg_ms"], "serial", e["serial_leg"], e["length_buckets"]["buckets"], e.get("host_ms_per_step"))
PY
