#!/bin/bash
# developer A/B of a run-time option with the contract bench, alternating: tools/ab_opt.sh "<bench args>" "<--set a=1>" "<--set a=0>" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${4:-3}); do for o in "$2" "$3"; do
  python3 $R/bench.py --no-cpu $1 $o 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$o]', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done; done
