#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_smalltrace
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
for mode in graph eager; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -- python3 $R/tools/enc_small_trace.py 50 $mode > $O/$mode.out 2>&1 || { tail -5 $O/$mode.out; exit 1; }
python3 - <<PY
import csv, glob
rows=list(csv.DictReader(open(glob.glob("$O/$mode/*/*_kernel_stats.csv")[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("$mode: kernel time per encode %.3f ms over 50 encodes" % (tot/50/1e6))
for r in rows[:12]:
    print(f"  {float(r['TotalDurationNs'])/tot*100:5.1f}% calls/encode {int(r['Calls'])/50:6.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:90]}")
# timeline of the last encode: span vs busy
t=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in csv.DictReader(open(glob.glob("$O/$mode/*/*_kernel_trace.csv")[0])))
last=t[-200:]
print("  last 200 kernels: span %.3f ms, busy %.3f ms" % ((last[-1][1]-last[0][0])/1e6, sum(b-a for a,b in last)/1e6))
PY
rm -rf $O/$mode
done
