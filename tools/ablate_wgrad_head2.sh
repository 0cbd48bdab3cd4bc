#!/bin/bash
# timing-only ablation of conv_wgrad_head2_kernel (results garbage): WH_ABL bits 1 no MFMAs, 2 no transposed operand reads, 4 no global loads, 8 no x staging stores
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
cp lib/libctseg_hip.so /tmp/lib_keep.so; cp build/conv_wgrad_halo.o /tmp/wh_keep.o
for d in 0 1 2 3 4 8 12 7 15; do
  /opt/rocm/bin/hipcc $FLAGS -DWH_ABL=$d -c csrc/conv_wgrad_halo.hip -o build/conv_wgrad_halo.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  echo -n "WH_ABL=$d "; (cd .. && timeout -k 5 120 python tools/bench_layers.py --only bwd:0 --loop 50 2>/dev/null | tail -1 | sed 's/.*avg/avg/')
done
cp /tmp/lib_keep.so lib/libctseg_hip.so; cp /tmp/wh_keep.o build/conv_wgrad_halo.o
