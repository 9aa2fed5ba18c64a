#!/bin/bash
# rocprofv3 kernel stats of one bench workload: tools/r03_trace.sh TAG "<bench args>"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py $2 > $O/bench.json 2> $O/bench.err || echo "trace failed"
f=$(ls $O/raw/*/*kernel_stats.csv | head -1)
python3 - "$f" "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows:
        if r[0] == "Name" or "rdx" in r[0]:
            w.writerow([r[0][:110]] + r[1:])
for r in rows:
    if r[0] == "Name" or "rdx" in r[0]:
        print(r[0][:70], r[1:6])
PY
rm -rf $O/raw
