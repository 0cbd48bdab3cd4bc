import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-image-segmentation_amd")):
    sys.path.insert(0, p)
from capstone_amd import _native as nat
dev = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
def bench(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, S, C, ld) in ((2, 24576, 256, 256), (2, 24576, 256, 512), (2, 24576, 128, 128), (2, 196608, 64, 64), (2, 1572864, 32, 32)):
    g = torch.randn(N, S, ld, device=dev).bfloat16(); y = torch.randn(N, S, ld, device=dev).bfloat16(); dy = torch.empty(N, S, C, device=dev).bfloat16()
    mr = torch.rand(N, C, 2, device=dev); al = torch.tensor([0.25], device=dev); sums = torch.rand(N, C, 2, device=dev)
    for rows_per in (512, 128, 48, 32):
        P = max(1, -(-S // rows_per)); P = min(P, 4096)
        part = torch.zeros(N, P, 3, C, device=dev)
        t = bench(lambda: nat.call("ctseg_instnorm_prelu_bwd_reduce", nat.BF16, g.data_ptr(), ld, y.data_ptr(), ld, mr.data_ptr(), al.data_ptr(), part.data_ptr(), P, C, N, S, C))
        print("reduce N%d S%d C%d ld%d rows_per %d P %d: %.1f us" % (N, S, C, ld, rows_per, P, t))
    dap = torch.zeros(N * C + 1, dtype=torch.float64, device=dev); dal = torch.zeros(1, device=dev)
    t = bench(lambda: nat.call("ctseg_instnorm_prelu_bwd_apply", nat.BF16, g.data_ptr(), ld, y.data_ptr(), ld, mr.data_ptr(), al.data_ptr(), sums.data_ptr(), dy.data_ptr(), C, None, 0, N, S, C, dap.data_ptr(), N * C, dal.data_ptr()))
    print("apply  N%d S%d C%d ld%d: %.1f us  (%.0f MB)" % (N, S, C, ld, t, N * S * C * 2 * 3 / 1e6))
