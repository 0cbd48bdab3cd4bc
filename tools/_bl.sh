python tools/bench_layers.py > gpurun_out/layers_r2_x1.txt 2>&1
CTSEG_NO_HALO_X=1 python tools/bench_layers.py > gpurun_out/layers_r2_x0.txt 2>&1
grep -E "Cg= 32 Cn= 32|sum of" gpurun_out/layers_r2_x1.txt gpurun_out/layers_r2_x0.txt
