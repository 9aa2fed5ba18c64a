#!/bin/bash
# round 4, first measurement of the single-question forward: new kernels' tests, the knob sweep, a kernel trace of the default.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_enc1
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_embedding_provider.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
RDX_ENC_OLD=1 timeout -k 10 200 python3 tools/enc_small_bench.py 2>/dev/null | tee $O/sweep.txt
for o in 16 8 4; do for f in 16 8 4; do for pf in 1 0; do
  RDX_ENC_FPB_O=$o RDX_ENC_FPB_F2=$f RDX_ENC_PREFETCH=$pf timeout -k 10 200 python3 tools/enc_small_bench.py 10 2>/dev/null | tee -a $O/sweep.txt
done; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/enc_small_bench.py 5 > $O/trace.out 2> $O/trace.err || echo "trace failed"
f=$(ls $O/trace/*/*kernel_stats.csv | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/trace
head -30 $O/kernel_stats.csv
