#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_graph
mkdir -p $O && cd $R
timeout -k 10 400 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -1 $O/pytest.txt
ENC_AB_GRAPH=1 timeout -k 10 400 python3 tools/enc_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/enc_ab.txt
for rnd in 1 2; do for v in graph eager; do
  if [ $v = graph ]; then F=--enc-graphs; else F=; fi
  timeout -k 10 400 python3 bench.py --workload c5 --no-cpu $F > $O/c5_${v}_$rnd.json 2> /dev/null || echo fail $v
  python3 -c "
import json; d=json.load(open('$O/c5_${v}_$rnd.json')); e=d['encode']
print('$v', $rnd, d['ms_per_step'], 'enc', e['avg_ms'], 'serial', e['serial_leg']['ms_per_step'], e['host_ms_per_step'])"
done; done
