#!/bin/bash
# developer A/B of whole-library variants with the contract bench: tools/ab_bench.sh "<variant> <variant> ..." [bench args]
# variant = "base" (rag_dpo_amd/librdx.so) or NAME (tools/librdx_NAME.so, built by tools/ab_lib.py build NAME "-D...")
# The product library is never overwritten: a variant is selected through RDX_LIB_PATH (rag_dpo_amd/_lib.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for v in $1; do
  if [ "$v" = base ]; then L=$R/rag_dpo_amd/librdx.so; else L=$R/tools/librdx_$v.so; fi
  RDX_LIB_PATH=$L python3 $R/bench.py --no-cpu $2 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done; done
