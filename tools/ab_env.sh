#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh VAR v0 v1 [steps]  ->  ms/step of bench.py with VAR=v0 and VAR=v1, twice each (interleaved)
VAR=$1; A=$2; B=$3; STEPS=${4:-100}
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps $STEPS --no-cpu-baseline --fp32-steps 0 --drop-in-steps 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$VAR=$v', round(d['ms_per_step'],4))"
  done
done
