#!/bin/bash
# same-box A/B of the ring-pipelined weight-gradient kernel: per-op table (weight-gradient rows) and ms/step with CTSEG_WGRAD_RING=0 / 1
#   tools/ab_wgrad_ring.sh <out dir under gpurun_out/>
O=$(pwd)/$1; mkdir -p "$O"
for v in 0 1; do
  CTSEG_WGRAD_RING=$v timeout -k 10 300 python tools/bench_layers.py --reps 5 > "$O/per_op_ring$v.txt" 2>&1
  grep "conv_wgrad\|sum of" "$O/per_op_ring$v.txt" | sed "s/^/ring=$v /"
done
tools/ab_env.sh CTSEG_WGRAD_RING 0 1 100 | tee "$O/step_ab.txt"
