#!/bin/bash
# rocprofv3 kernel stats of 20 bench steps -> gpurun_out/$1/prof/bench_kernel_stats.csv (run from the repo root on the GPU box)
TAG=${1:-prof}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --drop-in-steps 0 > $O/prof_bench.json 2> $O/prof.err
echo "prof rc=$?"
