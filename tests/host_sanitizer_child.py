"""Run INSIDE a process that preloaded the AddressSanitizer runtime and points CTSEG_LIB at the host-sanitizer build
(tests/test_host_sanitizer.py starts it): drives every host-only path of the C ABI — descriptor validation, kernel selection,
sizing / capability queries — over a few thousand random descriptors, plus the argument checks of the launch entry points that
reject before any HIP call.  No GPU, no launches.  Prints "SANITIZER-CHILD-OK <n>" when it got through."""
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-image-segmentation_amd"))
from capstone_amd import _native as nat  # noqa: E402

L = nat.lib()
assert "lib_asan" in nat.LIB_PATH, nat.LIB_PATH
rnd = random.Random(12342)
buf = ctypes.create_string_buffer(1 << 16)          # a real (host) address for the "pointer" fields: never dereferenced by the host code
base = ctypes.addressof(buf)


def tap(dx, dy, dz):
    return (dx & 255) | ((dy & 255) << 8) | ((dz & 255) << 16)


def rand_conv():
    d = nat.ConvDesc()
    d.dtype = rnd.choice([nat.F32, nat.BF16, nat.F16, 7])
    d.N = rnd.choice([1, 2, 3, 17])
    sp = [rnd.choice([1, 3, 4, 8, 16, 24, 48, 64, 256, 512]) for _ in range(3)]
    d.sin, d.sout = rnd.choice([(1, 1), (2, 1), (1, 2), (3, 1)])
    d.Xr, d.Yr, d.Zr = sp
    d.Xi, d.Yi, d.Zi = [s * d.sin for s in sp]
    d.Xo, d.Yo, d.Zo = [s * d.sout for s in sp]
    d.Cg = rnd.choice([1, 8, 16, 32, 64, 128, 256, 384])
    d.Cn = rnd.choice([10, 12, 16, 32, 64, 128, 256, 384])
    d.Cn_store = rnd.choice([d.Cn, (d.Cn + 7) // 8 * 8, (d.Cn + 3) // 4 * 4])
    d.g_ld = rnd.choice([d.Cg, 12, d.Cg + 8])
    d.o_ld = rnd.choice([d.Cn_store, 12, d.Cn_store + 8])
    d.add_ld = d.o_ld
    d.in_, d.w, d.out = base, base + 4096, base + 8192
    if rnd.random() < 0.4:
        d.add = rnd.choice([base, base + 12288])
    if rnd.random() < 0.3:
        d.out2, d.out2_col0, d.o2_ld = base + 16384, rnd.choice([0, 4, 16, 32, 64]), rnd.choice([16, 32, 64])
    d.out_f32, d.add_f32 = rnd.choice([0, 0, 1]), rnd.choice([0, 0, 1])
    d.nclass = rnd.choice([1, 1, 8, 9])
    for c in range(min(d.nclass, nat.MAX_CLASSES)):
        k = d.cls[c]
        k.ntaps = rnd.choice([1, 2, 4, 8, 27, 28])
        k.kpad = (max(1, min(k.ntaps, 27)) * d.Cg + 127) // 128 * 128 + rnd.choice([0, 0, 64])
        k.w_off = rnd.choice([0, 8, 1024, 3])
        k.ox, k.oy, k.oz = [rnd.choice([0, 1]) for _ in range(3)]
        order = rnd.choice([1, -1])
        for j in range(min(k.ntaps, 27)):
            k.taps[j] = tap(order * (j // 9 - 1), order * ((j // 3) % 3 - 1), order * (j % 3 - 1)) if k.ntaps == 27 else tap(0, 0, rnd.choice([-1, 0, 1, 2]))
    if rnd.random() < 0.3:
        d.in_mean_rstd, d.in_alpha, d.in_norm_C = base, base, rnd.choice([10, 12, 32])
    if rnd.random() < 0.4:
        d.bst_y, d.bst_y_ld, d.bst_C, d.bst_col0 = base + 20480, rnd.choice([12, 16, 32, 64, 256]), rnd.choice([10, 32, 64, 256]), rnd.choice([0, 0, 32, 128, 5])
    return d


def rand_wgrad():
    d = nat.WgradDesc()
    d.dtype = rnd.choice([nat.F32, nat.BF16, nat.F16])
    d.N = rnd.choice([1, 2])
    sp = [rnd.choice([3, 4, 8, 16, 24, 64, 256]) for _ in range(3)]
    d.sin = rnd.choice([1, 2])
    d.Xr, d.Yr, d.Zr = sp
    d.Xi, d.Yi, d.Zi = [s * d.sin for s in sp]
    d.Cg, d.Cn = rnd.choice([1, 16, 32, 64, 256]), rnd.choice([10, 16, 64, 256, 384])
    d.g_ld, d.d_ld = rnd.choice([d.Cg, 12]), rnd.choice([(d.Cn + 7) // 8 * 8, 12])
    d.ntaps = rnd.choice([1, 27])
    for j in range(d.ntaps):
        d.taps[j] = tap(j // 9 - 1, (j // 3) % 3 - 1, j % 3 - 1)
    d.splits = rnd.choice([1, 8, 72, 128])
    d.kpad_w = (d.ntaps * d.Cg + 1 + 127) // 128 * 128
    d.cn_pad = (d.Cn + 127) // 128 * 128
    d.in_, d.dy, d.ws = base, base + 4096, base + 8192
    return d


n = 0
for _ in range(4000):
    d = rand_conv()
    r = ctypes.byref(d)
    for q in ("ctseg_conv_num_tiles", "ctseg_conv_split_ok", "ctseg_conv_narrow_ok", "ctseg_conv_in_norm_ok", "ctseg_conv_bwd_stats_slots"):
        getattr(L, q)(r)
        n += 1
    L.ctseg_conv_logits_ce_slots(r, rnd.choice([2, 10, 12, 13]))
    # launch entry points with arguments that are rejected BEFORE any HIP call: a foreign struct size, null tensors
    d.struct_size -= 8
    assert L.ctseg_conv_igemm(r, None) < 0
    d.struct_size += 8
    d.out = None
    assert L.ctseg_conv_igemm(r, None) < 0 and L.ctseg_last_error()
    n += 3
for _ in range(2000):
    w = rand_wgrad()
    r = ctypes.byref(w)
    L.ctseg_conv_wgrad_slabs(r)
    L.ctseg_wgrad_narrow_ok(r)
    L.ctseg_wgrad_in_norm_ok(r)
    w.ws = None
    assert L.ctseg_conv_wgrad(r, None) < 0
    n += 4
for c in (1, 10, 16, 33, 64, 200, 256, 1024):
    L.ctseg_conv_tile_rows(c), L.ctseg_conv_tile_cols(c), L.ctseg_wgrad_tile_cols(c)
assert L.ctseg_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) < 0
print("SANITIZER-CHILD-OK", n)
