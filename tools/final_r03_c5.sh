#!/bin/bash
# The config-5 lines of tools/final_r03.sh alone (the encoder changed after that call: the attention's LDS window is sized from the
# longest text): bench lines, the encode's GPU time behind a plug, its kernel stats and the PMC traffic of the two librdx kernels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r03
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu > $O/c5.json 2> $O/c5.err || echo "bench c5 failed"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu --enc-torch-ops > $O/c5_torch_ops.json 2> /dev/null || echo "bench c5 torch ops failed"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu --enc-module-forward > $O/c5_module_forward.json 2> /dev/null || echo "bench c5 module failed"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu > $O/c5_again.json 2> /dev/null || echo "bench c5 again failed"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu --enc-graphs > $O/c5_graph_replay.json 2> /dev/null || echo "bench c5 graphs failed"
ENC_AB_GRAPH=1 timeout -k 10 300 python3 tools/enc_ab.py 2>&1 | grep -v amdgpu.ids > $O/c5_encode_gpu_time.txt || echo "enc_ab failed"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_enc -- python3 $R/tools/enc_fused_only.py 20 > /dev/null 2> $O/trace_enc.err || echo "trace enc failed"
f=$(ls $O/trace_enc/*/*kernel_stats.csv | head -1); python3 - "$f" "$O/c5_encode_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows[:16]:
        w.writerow([r[0][:140]] + r[1:])
PY
rm -rf $O/trace_enc
cd $R
bash tools/r03_encpmc.sh > /dev/null 2>&1 && cp $R/gpurun_out/r03_encpmc/c5_encode_traffic_pmc.txt $O/ || echo "enc pmc failed"
python3 - <<PY
import json
for n in ("c5", "c5_again", "c5_graph_replay", "c5_torch_ops", "c5_module_forward"):
    d = json.load(open("$O/%s.json" % n)); e = d["encode"]
    print(n, d["ms_per_step"], "enc", e["avg_ms"], "serial", e["serial_leg"]["ms_per_step"], e["host_ms_per_step"], "scan", d["roofline"]["avg_launch_ms"])
PY
cat $O/c5_encode_gpu_time.txt | tail -4; cat $O/c5_encode_traffic_pmc.txt | tail -2; head -7 $O/c5_encode_kernel_stats.csv | cut -c1-60,140-260
