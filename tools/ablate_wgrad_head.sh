#!/bin/bash
# timing-only ablation of conv_wgrad_head_kernel on the GPU box (WH_ABL bits: 1 no MFMAs, 2 no LDS operand reads, 4 no global loads,
# 8 no LDS staging stores).  usage: bash tools/ablate_wgrad_head.sh > gpurun_out/abl_wh.txt
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in 0 1 2 3 4 8 12 15 0; do
  /opt/rocm/bin/hipcc $FLAGS -DWH_ABL=$d -c csrc/conv_wgrad_halo.hip -o build/conv_wgrad_halo.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  echo -n "WH_ABL=$d "; (cd .. && python tools/bench_layers.py --only bwd:17 --loop 20 2>/dev/null | tail -1 | sed 's/ctseg_conv_wgrad wgrad//')
done
