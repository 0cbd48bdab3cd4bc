#!/bin/bash
# HBM-side bytes per kernel over whole training steps (single stream, so counters are not mixed across concurrent kernels):
# one rocprofv3 --pmc pass for FETCH_SIZE, one for WRITE_SIZE; tools/pmc_step_summarise.py prints per-kernel-name averages.
set -e
OUT=${1:-gpurun_out/pmc_step}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && CTSEG_SIDE_STREAM=0 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/$grp" -- \
     python3 "$ROOT/bench.py" --no-cpu-baseline --fp32-steps 0 --steps 3 --warmup 2 > "$ROOT/$OUT/$grp.log" 2>&1)
done
python3 tools/pmc_step_summarise.py "$OUT"
