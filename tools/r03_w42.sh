#!/bin/bash
# Prices a 4 x 2 wave layout for the B = 1024 main scan without building it: ablation builds of the final kernel
# (tools/ab_lib.py build NAME "-D..."; timing only, scores are garbage), alternating on one box, then one PMC pass each.
#   noemit  the kernel without its emit path (what every ablation build is compared with)
#   halfb   + half the query-fragment ds_read_b128 (16 per wave and k-step)
#   dbla    + twice the corpus loads (8 per wave and k-step, every line requested by two waves of the CU)
#   w42     both = the instruction mix of a 4 x 2 layout, minus its 32 extra registers
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_w42
mkdir -p $O
cd $R
for round in 1 2; do
  for v in noemit halfb dbla w42; do
    timeout -k 10 240 python3 tools/ab_lib.py run $v 2000000 > $O/run_${v}_$round.txt 2>&1 || { echo "run $v failed"; tail -5 $O/run_${v}_$round.txt; exit 1; }
    grep '"b": 1024' $O/run_${v}_$round.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', $round, d['scan_ms'], d['scan_TFLOPs'])"
  done
done
for v in noemit halfb dbla w42; do
  RDX_LIB_PATH=$R/tools/librdx_$v.so timeout -k 10 300 bash tools/pmc.sh w42_$v "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" > $O/pmc_$v.txt 2>&1 || { echo "pmc $v failed"; tail -5 $O/pmc_$v.txt; exit 1; }
  grep "k_scan<256, 1" $O/pmc_$v.txt
  rm -rf $R/gpurun_out/pmc_w42_$v   # raw counter CSVs: tens of MB each
done
