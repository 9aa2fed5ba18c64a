#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c5b
mkdir -p $O; cd $R
python3 tools/enc_profile.py 2>&1 | grep -v amdgpu | sed -n 1,6p
for b in 4 1; do timeout -k 10 400 python3 bench.py --workload c5 --no-cpu --enc-buckets $b > $O/c5_b$b.json 2> $O/c5_b$b.err || echo fail $b; done
python3 - <<PY
import json
for b in (4,1):
    d=json.load(open("$O/c5_b%d.json"%b)); e=d["encode"]
    print(b, d["value"], d["ms_per_step"], "enc", e["avg_ms"], "serial", e["serial_leg"], e["length_buckets"]["buckets"])
PY
