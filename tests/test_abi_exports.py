"""CPU: libctseg_hip.so loads without a GPU and exports every symbol include/ctseg_hip.h declares; the ctypes
binding table covers exactly that set; argument validation rejects bad descriptors without touching a device."""
import ctypes
import os
import re

import pytest

from capstone_amd import _native as nat

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "ctseg_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctseg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(nat.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ctseg_hip.h but not exported"
    assert sorted(nat.EXPORTS) == names


def test_host_side_queries_and_validation():
    L = nat.lib()
    hdr = int(re.search(r"#define CTSEG_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert L.ctseg_abi_version() == hdr == nat.ABI_VERSION == 3
    assert (L.ctseg_conv_tile_rows(10), L.ctseg_conv_tile_rows(256)) == (256, 128)
    assert [L.ctseg_conv_tile_cols(c) for c in (10, 32, 64, 256)] == [16, 32, 64, 128]
    assert [L.ctseg_wgrad_tile_cols(c) for c in (10, 32, 64, 256)] == [16, 32, 64, 128]
    d = nat.ConvDesc()                      # all-zero descriptor: rejected before any launch
    assert L.ctseg_conv_igemm(ctypes.byref(d), None) < 0
    assert b"null pointer" in L.ctseg_last_error()
    with pytest.raises(nat.NativeError):
        nat.check(-1, "x")


def test_descriptors_carry_their_struct_size_and_a_foreign_size_is_rejected():
    """ABI 2: both descriptor structs start with the sizeof() their caller was compiled against; a caller built against another
    header (round 2 appended fields under version 1) is refused by every entry point instead of being read past its end."""
    L = nat.lib()
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for cname, cls in (("ctseg_conv_desc", nat.ConvDesc), ("ctseg_wgrad_desc", nat.WgradDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        first = [ln.strip() for ln in body.splitlines() if ln.strip()][0]
        assert first.startswith("int32_t struct_size"), first
        assert cls._fields_[0][0] == "struct_size" and cls().struct_size == ctypes.sizeof(cls)
        # every header field is mirrored, in order (names: `in` is spelled in_ in Python)
        names = re.findall(r"(?:\*|\s)([A-Za-z_][A-Za-z0-9_]*)(?:\[[A-Z_]+\])?\s*[,;]", body)
        assert [n if n != "in" else "in_" for n in names] == [f[0] for f in cls._fields_], (cname, names)
    d = nat.ConvDesc()
    d.struct_size -= 24                      # "compiled against the round-1 header"
    assert L.ctseg_conv_igemm(ctypes.byref(d), None) < 0 and b"struct_size" in L.ctseg_last_error()
    assert L.ctseg_conv_num_tiles(ctypes.byref(d)) < 0
    assert L.ctseg_conv_split_ok(ctypes.byref(d)) == 0 and L.ctseg_conv_narrow_ok(ctypes.byref(d)) == 0
    assert L.ctseg_conv_in_norm_ok(ctypes.byref(d)) == 0 and L.ctseg_conv_logits_ce_slots(ctypes.byref(d), 10) == 0
    w = nat.WgradDesc()
    w.struct_size += 8
    assert L.ctseg_conv_wgrad(ctypes.byref(w), None) < 0 and b"struct_size" in L.ctseg_last_error()
    assert L.ctseg_conv_wgrad_slabs(ctypes.byref(w)) < 0
    assert L.ctseg_wgrad_narrow_ok(ctypes.byref(w)) == 0 and L.ctseg_wgrad_in_norm_ok(ctypes.byref(w)) == 0


def test_product_has_no_cpu_fallback():
    import torch
    from capstone_amd.models import UNet
    net = UNet(3, 1, 10, (4, 8), (2,), num_res_units=2)
    with pytest.raises(nat.NativeError):
        net(torch.zeros(1, 1, 8, 8, 8))     # CPU tensor: loud failure, not an eager path
    with pytest.raises(nat.NativeError):
        net.model[0].conv.unit0(torch.zeros(1, 1, 8, 8, 8))   # containers never compute


def test_pack_block_structs_mirror_the_header():
    """ctseg_pack_block / ctseg_pack_part (ABI 3, ctseg_pack_weights): field order, array sizes and the struct size of the ctypes
    mirror follow the header (natural alignment on both sides)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for cname, cls in (("ctseg_pack_part", nat.PackPart), ("ctseg_pack_block", nat.PackBlock)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        names = re.findall(r"(?:\*|\s)([A-Za-z_][A-Za-z0-9_]*)(?:\[[A-Za-z_0-9]+\])?\s*[,;]", body)
        assert names == [f[0] for f in cls._fields_], (cname, names)
    assert ctypes.sizeof(nat.PackPart) == 32 and ctypes.sizeof(nat.PackBlock) == 8 + 4 * 4 + 8 + 2 * 32 + 4 * nat.MAX_TAPS + 4
    assert int(re.search(r"#define CTSEG_PACK_LDS_FLOATS (\d+)", open(HEADER).read()).group(1)) == nat.PACK_LDS_FLOATS


def test_weight_gradient_sizing_query_and_the_split_model():
    """ctseg_conv_wgrad_wgs_per_slab (host-side, no GPU): which kernel a descriptor gets and what one slab of it costs; and the split
    count capstone_amd.engine.GemmLayer._wgrad_splits derives from it for the many-channel layers of the reference's network
    (UNet(3,1,10,(32,64,128,256),(2,2,2,2),2) on 2x1x512x512x48, /root/reference/capstone/volumetric/base_trainer.py:65-72)."""
    import types
    from capstone_amd.engine import GemmLayer
    L = nat.lib()

    def desc(cg, cn, rows, sin=1, dt=nat.BF16, taps=27, N=2):
        d = nat.WgradDesc()
        d.dtype = dt
        d.in_, d.dy = 4096, 8192                       # (only their alignment is looked at)
        d.N = N
        d.Xr, d.Yr, d.Zr = rows
        d.Xi, d.Yi, d.Zi = [r * sin for r in rows]
        d.Cg, d.Cn, d.g_ld, d.d_ld, d.sin, d.ntaps = cg, cn, cg, cn, sin, taps
        bnw = L.ctseg_wgrad_tile_cols(cn)
        d.splits, d.kpad_w, d.cn_pad = 1, -(-(taps * cg + 1) // 128) * 128, -(-cn // bnw) * bnw
        return d

    def ask(d):
        pc, sb = ctypes.c_int32(-1), ctypes.c_int32(-1)
        return L.ctseg_conv_wgrad_wgs_per_slab(ctypes.byref(d), ctypes.byref(pc), ctypes.byref(sb)), pc.value, sb.value

    # ring kernel: 256 x 256 tile (cn_pad % 256 == 0), 256 x 128, 512 x 64; one workgroup per CU; bytes of a 32-row stage
    assert ask(desc(256, 256, (64, 64, 6))) == (28, 1, 32 * 512 * 2)          # 27 K tiles + the bias row's own
    assert ask(desc(128, 128, (64, 64, 6))) == (14, 1, 32 * 384 * 2)
    assert ask(desc(64, 384, (64, 64, 6), sin=2)) == (7 * 3, 1, 32 * 384 * 2)
    assert ask(desc(64, 64, (128, 128, 12))) == (4, 1, 32 * 576 * 2)
    # fp32 storage and the few-channel layers keep the generic kernel (four 128-row tiles per CU) ...
    assert ask(desc(64, 64, (128, 128, 12), dt=nat.F32)) == (14, 4, 32 * (128 + 64) * 4)
    assert ask(desc(48, 40, (40, 40, 20)))[1] == 4
    # ... persistent LDS-halo kernels size their own grid
    assert ask(desc(32, 32, (256, 256, 24))) == (0, 0, 0)
    z = nat.WgradDesc()
    z.struct_size = 8
    assert L.ctseg_conv_wgrad_wgs_per_slab(ctypes.byref(z), None, None) < 0

    stub = types.SimpleNamespace(plan=types.SimpleNamespace(dt=nat.BF16))
    for cg, cn, rows, sin in ((256, 256, (64, 64, 6), 1), (128, 256, (64, 64, 6), 1), (64, 64, (128, 128, 12), 1), (32, 128, (128, 128, 12), 2)):
        d = desc(cg, cn, rows, sin)
        s = GemmLayer._wgrad_splits(stub, d, 2, rows[0] * rows[1] * rows[2], False)
        wps = ask(d)[0]
        assert 1 <= s and rows[0] * rows[1] * rows[2] // s >= 192, (cg, cn, s)
        assert 192 <= wps * 2 * s <= 256, (cg, cn, s, wps * 2 * s)           # one nearly full round of 256 workgroups
    # the generic rule is untouched (fp32 storage: 2048 workgroups aimed for, slabs in multiples of 8)
    stub32 = types.SimpleNamespace(plan=types.SimpleNamespace(dt=nat.F32))
    assert GemmLayer._wgrad_splits(stub32, desc(64, 64, (128, 128, 12), dt=nat.F32), 2, 128 * 128 * 12, False) == 72
