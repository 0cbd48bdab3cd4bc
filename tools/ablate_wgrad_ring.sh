#!/bin/bash
# timing-only ablation of conv_wgrad_ring_kernel (results garbage): WR_ABL bits 1 no MFMAs, 2 no transposed fragment reads, 4 no gathered-operand
# loads, 8 no dy loads, 16 no step barrier, 32 no bounds tests / coordinate carries.   tools/ablate_wgrad_ring.sh [bench_layers op, default bwd:24 = 256->256]
OP=${1:-bwd:24}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
cp lib/libctseg_hip.so /tmp/lib_keep.so; cp build/conv_wgrad_ring.o /tmp/wr_keep.o
for d in ${ABLS:-0 1 2 3 4 8 12 15 16 19 31 32 63}; do
  /opt/rocm/bin/hipcc $FLAGS -DWR_ABL=$d -c csrc/conv_wgrad_ring.hip -o build/conv_wgrad_ring.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  echo -n "WR_ABL=$d "; (cd .. && timeout -k 5 120 python tools/bench_layers.py --only $OP --loop 50 2>/dev/null | tail -1 | sed 's/.*avg/avg/')
done
cp /tmp/lib_keep.so lib/libctseg_hip.so; cp /tmp/wr_keep.o build/conv_wgrad_ring.o
