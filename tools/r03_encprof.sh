#!/bin/bash
# kernel-level breakdown of the packed encoder forward (1024 short texts): rocprofv3 --kernel-trace --stats over tools/enc_profile.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_encprof
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/enc_fused_only.py 20 > $O/out.txt 2>&1 || { tail -20 $O/out.txt; exit 1; }
tail -1 $O/out.txt
f=$(ls $O/prof/*/*_kernel_stats.csv | head -1)
cp $f $O/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
find $O/prof -size +1M -delete
