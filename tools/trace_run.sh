#!/bin/bash
# rocprofv3 kernel trace of a few bench steps -> per-queue timeline of the last full step: tools/trace_run.sh <outdir> [ENV=VAL ...]
# (the environment assignments are exported BEFORE rocprofv3 starts: never put env / bash -c between rocprofv3's "--" and python3)
O=$(pwd)/$1; shift
for kv in "$@"; do export "$kv"; done
R=$(pwd)
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --fp32-steps 0 --drop-in-steps 0 > $O/bench.json 2> $O/prof.err)
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T 40 > $O/timeline_by_queue.txt 2>&1
python tools/trace_step.py $T > $O/timeline_full.txt 2>&1
rm -rf $O/prof
