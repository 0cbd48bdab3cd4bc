#!/bin/bash
# timing-only ablation of the fused head (conv_halo_x.hip, CE variant): X_ABL bits 1 no MFMAs, 4 no global loads of the halo, 8 no loss arithmetic
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in 0 16 32 64 48 112; do
  /opt/rocm/bin/hipcc $FLAGS -DX_ABL=$d -c csrc/conv_halo_x.hip -o build/conv_halo_x.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  echo -n "X_ABL=$d "; (cd .. && python tools/time_head_ce.py 2>/dev/null | tail -1)
done
