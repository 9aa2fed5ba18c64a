#!/bin/bash
# round 4: fuzz hunts on the final library (seeds of their own; wave_layout is among the drawn options)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_fuzz
mkdir -p $O
cd $R
for seed in 4302 4303 4304 4305; do
  RDX_FUZZ_SEED=$seed RDX_FUZZ_CASES=250 timeout -k 10 280 python3 -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz_$seed.log 2>&1; echo "fuzz $seed rc $?" | tee -a $O/summary.txt
  tail -1 $O/fuzz_$seed.log | tee -a $O/summary.txt
done
