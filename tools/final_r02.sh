#!/bin/bash
# One gpurun call: the judged measurements of the round (bench lines of every BASELINE config, rocprofv3 kernel stats of
# the dominant kernel of each, FETCH/WRITE PMC passes for c4, one SQ/GRBM pass). Output under gpurun_out/final_r02 + prof_r02.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r02
mkdir -p $O
cd $R
for w in c4 c3 c2 c1; do python3 bench.py --workload $w > $O/$w.json 2> $O/$w.err || echo "bench $w failed"; done
python3 bench.py --workload c5 --no-cpu > $O/c5.json 2> $O/c5.err || echo "bench c5 failed"
python3 bench.py --workload c5 --no-cpu --no-encode > $O/c5_noenc.json 2> /dev/null || echo "bench c5 noenc failed"
python3 bench.py --workload c4 --no-cpu --set fuse_epilogue=0 > $O/c4_unfused.json 2> /dev/null || echo "bench c4 unfused failed"
python3 bench.py --workload c4 --no-cpu --set sib_sync=1 > $O/c4_sibsync.json 2> /dev/null || echo "bench c4 sib failed"
bash tools/prof.sh r02 "--steps 5 --warmup 2 --no-cpu" || echo "prof failed $?"
cd /tmp && export TMPDIR=/tmp
for w in c1 c2 c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu > /dev/null 2> $O/trace_$w.err || echo "trace $w failed"
  f=$(ls $O/trace_$w/*/*kernel_stats.csv | head -1); python3 - "$f" "$O/${w}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows:
        if r[0] == "Name" or "rdx" in r[0] or "rocclr" in r[0]:
            w.writerow([r[0][:140]] + r[1:])
PY
  rm -rf $O/trace_$w
done
cd $R
bash tools/pmc.sh r02sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" > $O/sq_grbm_2Mrows_pmc.txt 2>&1 || echo "pmc failed"
rm -rf $R/gpurun_out/pmc_r02sq $R/gpurun_out/prof_r02/*/*/*kernel_trace.csv
python3 bench.py --workload c4 --no-cpu --rows 1250000 --steps 30 --warmup 5 > $O/c4_one_eighth.json 2> /dev/null || echo "one-eighth failed"
python3 bench.py --workload c4 --no-cpu --rows 1250000 --steps 30 --warmup 5 --force-dist > $O/c4_one_eighth_rccl_world1.json 2> /dev/null || echo "one-eighth rccl failed"
python3 tools/collection_latency.py 2>/dev/null | grep -v amdgpu > $O/collection_latency.txt
RDX_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --rows 600000 --steps 3 --warmup 1 --check-merged > $O/rehearse2_selflaunch.json 2> $O/rehearse2.err; echo "self-launch rehearsal rc=$?"
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"] and d["roofline"]["frac"], d.get("merged_equals_single_index"))
    except Exception as e:
        print(f, "ERR", e)
PY
