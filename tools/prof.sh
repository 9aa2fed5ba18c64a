#!/bin/bash
# rocprofv3 passes for the default bench (run on the GPU box via gpurun): kernel trace + stats, then PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_${1:-r01}
ARGS=${2:---steps 5 --warmup 2 --no-cpu}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err || exit 3
find $OUT -name "*.csv" | head -20
