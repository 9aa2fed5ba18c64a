#!/bin/bash
# round 4: VERDICT r3 item 6 — the one-wave-per-SIMD main scan (csrc/scan_w4.hpp, option wave_layout = 1) beside the shipped kernel on ONE box:
# TFLOP/s by librdx's HIP events (two alternating rounds, 10 M rows and a 2 M-row shard), three tuning variants, one PMC pass each for
# clock and MFMA busy. -> profiles/r04/c4_wave_layout_1x.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_w4
mkdir -p $O
cd $R
line() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"{sys.argv[2]:28s} {d['value']:9.1f} q/s  {d['ms_per_step']:7.3f} ms/step  main scan {r['avg_launch_ms']:7.3f} ms = {r['achieved']:7.1f} TFLOP/s  frac {r['frac']:.4f}  exact {((d.get('recall_at_10') or {}).get('ids_bit_exact'))}")
PY
}
for rep in 1 2; do
  for v in base0 base1 w4burst w4pd3 w4burstpd3; do
    case $v in
      base0) lib=""; wl=0;;
      base1) lib=""; wl=1;;
      *) lib=$R/tools/librdx_$v.so; wl=1;;
    esac
    RDX_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --workload c4 --rows 2000000 --no-others --no-cpu --steps 60 --warmup 15 --set wave_layout=$wl > $O/${v}_2M_$rep.json 2> $O/${v}_2M_$rep.err || echo "$v failed"
    line $O/${v}_2M_$rep.json "2M rows $v rep $rep" | tee -a $O/summary.txt
  done
done
for v in base0 base1; do
  wl=${v#base}
  timeout -k 10 400 bash tools/pmc.sh r04w4_$v "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "--steps 3 --warmup 1 --no-cpu --no-others --rows 2000000 --set wave_layout=$wl" > $O/pmc_$v.txt 2>&1 || echo "pmc $v failed"
  cat $O/pmc_$v.txt | tee -a $O/summary.txt
  rm -rf $R/gpurun_out/pmc_r04w4_$v
done
