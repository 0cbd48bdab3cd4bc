#!/bin/bash
# same-box A/B of ONE kernel source on the GPU box: builds csrc/<name>.hip.base (a copy of the committed version, made before
# the call: git show HEAD:.../csrc/<name>.hip > .../csrc/<name>.hip.base) and the working-tree file in turn, times the given ops
# usage: tools/ab_kernel_file.sh conv_stem "fwd:0 bwd:83"
NAME=$1; OPS=${2:-"fwd:0"}
cd ct-image-segmentation_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wno-unused-result -fno-gpu-rdc"
for v in base new base new; do
  if [ $v = base ]; then SRC=csrc/$NAME.hip.base; else SRC=csrc/$NAME.hip; fi
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC -o build/$NAME.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libctseg_hip.so build/*.o
  for op in $OPS; do
    echo -n "$v "; (cd .. && python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1 | sed "s/in=.*avg/avg/")
  done
done
