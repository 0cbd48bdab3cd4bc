#!/bin/bash
# timing-only ablation of conv_up_halo.hip on the GPU box: rebuilds the one object with -DUP_ABL=<bits> (1 no MFMAs, 2 no LDS operand
# reads, 8 no output stores, 16 no halo loads), relinks, replays one recorded pass.
# usage: bash tools/ablate_up_halo.sh "fwd:37" > gpurun_out/abl_up.txt
OPS=${1:-"fwd:37"}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in 0 1 2 3 8 16 11 27 0; do
  /opt/rocm/bin/hipcc $FLAGS -DUP_ABL=$d -c csrc/conv_up_halo.hip -o build/conv_up_halo.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  for op in $OPS; do
    echo -n "UP_ABL=$d "; (cd .. && python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1 | sed "s/in=.*avg/avg/")
  done
done
