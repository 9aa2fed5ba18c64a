#!/bin/bash
# A/B: non-temporal corpus loads at B = 1024 (four query-tile workgroups share every corpus tile through L2): speed and fabric traffic
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_nt
mkdir -p $O && cd $R
for rnd in 1 2 3; do for v in base ntall; do
  if [ $v = base ]; then L=; else L=$R/tools/librdx_$v.so; fi
  RDX_LIB_PATH=$L timeout -k 10 300 python3 bench.py --no-cpu --steps 12 --warmup 4 > $O/c4_${v}_$rnd.json 2>/dev/null || echo fail $v
  python3 -c "
import json; d=json.load(open('$O/c4_${v}_$rnd.json')); print('$v', $rnd, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done; done
for v in base ntall; do
  if [ $v = base ]; then L=; else L=$R/tools/librdx_$v.so; fi
  RDX_LIB_PATH=$L timeout -k 10 300 bash tools/pmc.sh nt_$v "FETCH_SIZE" > $O/pmc_$v.txt 2>&1 || { echo "pmc $v failed"; tail -3 $O/pmc_$v.txt; }
  grep "k_scan<256, 1" $O/pmc_$v.txt; rm -rf $R/gpurun_out/pmc_nt_$v
done
