for op in fwd:3 bwd:79; do
for d in 0 1 2 4 6 7 5 3; do
echo -n "dbg=$d "; CTSEG_DBG=$d python tools/bench_layers.py --only $op --loop 30 2>/dev/null | tail -1 | sed 's/ctseg_conv_igemm conv Cg= 32 Cn= 32 in=256x256x24 rows=256x256x24//'
done; done
