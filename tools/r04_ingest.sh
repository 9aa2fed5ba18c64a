#!/bin/bash
# round 4: the single-question forward after the LDS-staged stage kernel, and the ingest measurement
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_ingest
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_embedding_provider.py tests/test_gpu_cabi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
RDX_ENC_OLD=1 timeout -k 10 200 python3 tools/enc_small_bench.py 10 2>/dev/null | tee $O/enc_small_sweep.txt
for o in 16 8 4; do for f in 16 8 4; do
  RDX_ENC_FPB_O=$o RDX_ENC_FPB_F2=$f timeout -k 10 200 python3 tools/enc_small_bench.py 10 2>/dev/null | tee -a $O/enc_small_sweep.txt
done; done
for a in torch mfma; do
  RDX_ENC_LONG_ATTN=$a timeout -k 10 500 python3 tools/ingest_bench.py 2>/dev/null | tee -a $O/ingest.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_q -- python3 $R/tools/enc_small_bench.py 5 > $O/trace_q.out 2> $O/trace_q.err || echo "trace failed"
f=$(ls $O/trace_q/*/*kernel_stats.csv | head -1); cp "$f" $O/enc_small_kernel_stats.csv; rm -rf $O/trace_q
for a in torch mfma; do
  RDX_ENC_LONG_ATTN=$a rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$a -- python3 $R/tools/ingest_bench.py 3000 > $O/trace_$a.out 2> $O/trace_$a.err || echo "trace $a failed"
  f=$(ls $O/trace_$a/*/*kernel_stats.csv | head -1); cp "$f" $O/ingest_${a}_kernel_stats.csv; rm -rf $O/trace_$a
  python3 $R/tools/classify_kernels.py $O/ingest_${a}_kernel_stats.csv | tee $O/ingest_${a}_split.json
done
