#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other .so> [reps] [steps] [bench_layers op]  -> ms/step with the in-tree build and with the other
OTHER=$(pwd)/$1; REPS=${2:-3}; STEPS=${3:-100}; OP=$4
for rep in $(seq 1 $REPS); do
  for lib in "" "$OTHER"; do
    if [ -n "$OP" ]; then echo -n "lib=${lib:-in-tree} "; CTSEG_LIB=$lib timeout -k 10 200 python tools/bench_layers.py --only $OP --loop 100 2>/dev/null | tail -1 | sed 's/.*avg/avg/'; fi
    echo -n "lib=${lib:-in-tree} "; CTSEG_LIB=$lib timeout -k 10 200 python bench.py --steps $STEPS --no-cpu-baseline --fp32-steps 0 --drop-in-steps 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],4))"
  done
done
