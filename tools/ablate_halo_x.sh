#!/bin/bash
# timing-only ablation of conv_halo_x.hip on the GPU box: rebuilds the one object with -DX_ABL=<bits> (1 no MFMAs, 2 no stores,
# 4 no DMA), relinks, replays one recorded pass.  usage: bash tools/ablate_halo_x.sh "fwd:3 bwd:79" > gpurun_out/abl.txt
OPS=${1:-"fwd:3"}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in 0 1 2 4 6 7 5 3 0; do
  /opt/rocm/bin/hipcc $FLAGS -DX_ABL=$d -c csrc/conv_halo_x.hip -o build/conv_halo_x.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  for op in $OPS; do
    echo -n "X_ABL=$d "; (cd .. && python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1)
  done
done
