#!/bin/bash
# timing-only ablation of conv_stem.hip's forward kernel on the GPU box
# (-DSTEM_ABL bits: 1 no MFMAs, 2 no LDS operand reads, 4 no patch loads, 8 no stores, 16 no statistics)
OPS=${1:-"fwd:0"}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in 0 1 2 4 8 16 24 28 31 0; do
  /opt/rocm/bin/hipcc $FLAGS -DSTEM_ABL=$d -c csrc/conv_stem.hip -o build/conv_stem.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  for op in $OPS; do
    echo -n "STEM_ABL=$d "; (cd .. && python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1 | sed "s/in=.*avg/avg/")
  done
done
