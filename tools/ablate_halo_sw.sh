#!/bin/bash
# timing-only ablation of conv_halo_sw.hip on the GPU box: rebuilds the one object with -DSW_ABL=<bits> (1 no MFMAs, 2 no LDS operand
# reads, 4 no weight stream, 8 no epilogue, 16 no halo loads), relinks, replays one recorded pass.
# usage: bash tools/ablate_halo_sw.sh "fwd:31 bwd:73" > gpurun_out/abl_sw.txt
OPS=${1:-"fwd:31"}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for d in ${ABLS:-0 8 32 15 31 63 0}; do
  /opt/rocm/bin/hipcc $FLAGS -DSW_ABL=$d -c csrc/conv_halo_sw.hip -o build/conv_halo_sw.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  for op in $OPS; do
    echo -n "SW_ABL=$d "; (cd .. && python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1)
  done
done
