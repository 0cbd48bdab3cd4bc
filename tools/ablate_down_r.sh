#!/bin/bash
# timing-only ablation of conv_down_r.hip on the GPU box (-DDR_ABL bits: 1 no MFMAs, 2 no LDS operand reads, 4 no DMA, 8 no stores)
OPS=${1:-"fwd:6"}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in 0 1 2 3 4 8 12 15 0; do
  /opt/rocm/bin/hipcc $FLAGS -DDR_ABL=$d -c csrc/conv_down_r.hip -o build/conv_down_r.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  for op in $OPS; do
    echo -n "DR_ABL=$d "; (cd .. && python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1 | sed "s/in=.*avg/avg/")
  done
done
