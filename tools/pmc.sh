#!/bin/bash
# rocprofv3 PMC pass (own run, no tracing flags) over a bench command; usage: tools/pmc.sh <tag> "<counters>" [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$1
ARGS=${3:---steps 3 --warmup 1 --no-cpu --rows 2000000}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --output-format csv -d $OUT -- python3 $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'k_scan' in r['Kernel_Name']:
        agg[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[r['Kernel_Name'][:40]]['_dur_ns'].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
