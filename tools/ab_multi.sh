#!/bin/bash
# interleaved A/B/C... of one environment switch on the same box: tools/ab_multi.sh VAR reps steps v0 v1 v2 ...  ->  ms/step per value, per repetition
VAR=$1; REPS=$2; STEPS=$3; shift 3
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps $STEPS --no-cpu-baseline --fp32-steps 0 --drop-in-steps 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$VAR=$v', round(d['ms_per_step'],4))"
  done
done
