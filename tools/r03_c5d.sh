#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c5d
mkdir -p $O; cd $R
for rnd in 1 2 3; do
  timeout -k 10 400 python3 bench.py --workload c5 --no-cpu > $O/c5_$rnd.json 2> /dev/null || echo fail
  python3 -c "
import json; d=json.load(open('$O/c5_$rnd.json')); e=d['encode']
print($rnd, d['value'], d['ms_per_step'], 'enc', e['avg_ms'], 'serial', e['serial_leg'], e.get('host_ms_per_step'), 'scan', d['roofline']['avg_launch_ms'])"
done
