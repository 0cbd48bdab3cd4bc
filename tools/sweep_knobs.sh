#!/bin/bash
# ms/step of bench.py under each setting of the scheduling knobs (one box, 150 steps each): tools/sweep_knobs.sh
run() { env "$@" timeout -k 10 200 python bench.py --steps 150 --no-cpu-baseline --fp32-steps 0 --drop-in-steps 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$*', round(d['ms_per_step'],4))"; }
run X=0
run CTSEG_DEFER_HEAD_WGRAD=1
run CTSEG_DEFER_HEAD_WGRAD=2
run CTSEG_WGRAD_TARGET_WGS=512
run CTSEG_WGRAD_TARGET_WGS=2048
run CTSEG_WH_PER_CU=1
run CTSEG_WH_PER_CU=2
run CTSEG_WH_PER_CU=4
run CTSEG_WU_PER_CU=1
run CTSEG_WU_PER_CU=2
run X=0
