#!/bin/bash
# Round artefacts on the GPU box (run from the repo root): rocprofv3 kernel stats of the bench command, per-op table, PMC passes of
# the bottleneck launch, sliding-window timings.  Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
TAG=${1:-r03}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench done" 
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 > $O/prof_bench.json 2> $O/prof.err); echo "rocprof done"
python tools/bench_layers.py > $O/per_op_table.txt 2>&1; echo "layers done"
bash tools/pmc_bottleneck.sh gpurun_out/$TAG/pmc > $O/pmc.log 2>&1; echo "pmc done"
python tools/bench_infer.py --precision fp16 > $O/infer_fp16.json 2> $O/infer.err; python tools/bench_infer.py --precision bf16 > $O/infer_bf16.json 2>> $O/infer.err; echo "infer done"
ls $O
