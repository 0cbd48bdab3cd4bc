#!/bin/bash
# round 4: where the fused head's time goes (timing-only builds; results garbage).  bits: 1 no MFMA, 4 no halo loads, 8 no loss arithmetic,
# 128 no staging stores, 256 no tile barrier, 1024 no operand reads, 2048 no exchange
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
cp lib/libctseg_hip.so /tmp/lib_keep.so; cp build/conv_halo_x.o /tmp/halo_x_keep.o
for d in 0 1 8 9 256 1024 1025 2048 2056 4 132 1165 1421 3469; do
  /opt/rocm/bin/hipcc $FLAGS -DX_ABL=$d -c csrc/conv_halo_x.hip -o build/conv_halo_x.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  echo -n "X_ABL=$d "; (cd .. && timeout -k 5 120 python tools/time_head_ce.py 2>/dev/null | tail -1)
done
cp /tmp/lib_keep.so lib/libctseg_hip.so; cp /tmp/halo_x_keep.o build/conv_halo_x.o
