"""CPU: checks on the gfx950 machine code that actually ships in lib/libctseg_hip.so (disassembled from the library's own
offload bundles, no recompilation, no GPU).

  * the wide-store hazard of DESIGN.md section 3.2g "Hardware fact 1" (a VALU write into the data registers of a
    buffer_store_dwordx3/x4 with an SGPR soffset within two issue slots: LLVM does not pad it, gfx950 needs it);
  * scratch (register spills) per kernel: a ratchet — known spillers may not grow, no new kernel may spill.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_store_hazard as H  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(H.LLVM, "llvm-objdump")), reason="ROCm LLVM tools missing")


def test_scanner_sees_a_planted_hazard_and_respects_s_nop():
    planted = """
0000000000001000 <_ZN5ctseg4testEv>:
        v_mov_b32 v10, v1
        buffer_store_dwordx4 v[10:13], v5, s[8:11], s46 offen
        v_mov_b32 v10, v2
        s_endpgm
"""
    wide, hz = H.scan_text(planted)
    assert wide == 1 and len(hz) == 1 and "v_mov_b32 v10" in hz[0]
    # second slot
    wide, hz = H.scan_text(planted.replace("        v_mov_b32 v10, v2", "        s_add_u32 s1, s1, 4\n        v_pk_add_f32 v[12:13], v[2:3], v[4:5]"))
    assert wide == 1 and len(hz) == 1
    # padded with s_nop: safe
    assert H.scan_text(planted.replace("        v_mov_b32 v10, v2", "        s_nop 1\n        v_mov_b32 v10, v2"))[1] == []
    # a write to OTHER registers, a store without an SGPR soffset (LLVM pads those itself), a 64-bit store: not this hazard
    assert H.scan_text(planted.replace("v_mov_b32 v10, v2", "v_mov_b32 v14, v2"))[1] == []
    assert H.scan_text(planted.replace("s46 offen", "0 offen")) == (0, [])
    assert H.scan_text(planted.replace("dwordx4 v[10:13]", "dwordx2 v[10:11]")) == (0, [])
    # third slot: outside the window
    assert H.scan_text(planted.replace("        v_mov_b32 v10, v2", "        s_mov_b32 s1, 0\n        s_mov_b32 s2, 0\n        v_mov_b32 v10, v2"))[1] == []


def test_no_wide_store_hazard_in_the_shipped_library():
    r = H.scan_library()
    assert r["bundles"] >= 15 and r["instructions"] > 100000           # every translation unit was found and disassembled
    assert r["wide_sgpr_stores"] >= 100                                   # ... and the stores in question exist (544 in round 2)
    assert r["hazards"] == [], "\n".join(r["hazards"])


def test_hand_waited_lds_reads_of_the_ring_weight_gradient():
    """conv_wgrad_ring.hip reads its MFMA fragments with inline-assembly ds_read_b64_tr_b16 (hipcc would otherwise drain the ring of
    direct-to-LDS loads in front of every read); nobody but the kernel's own `s_waitcnt lgkmcnt(0)` stands between such a read and
    the instruction that consumes its registers.  The scanner is checked on a planted violation, then run over the shipped code."""
    planted = """
0000000000001000 <_ZN5ctseg22conv_wgrad_ring_kernelI1EEv>:
        s_waitcnt vmcnt(3)
        s_barrier
        ds_read_b64_tr_b16 v[10:11], v5
        ds_read_b64_tr_b16 v[12:13], v5 offset:4096
        buffer_load_dwordx4 v7, s[20:23], 0 offen lds
        v_mfma_f32_16x16x32_bf16 a[0:3], v[20:23], v[24:27], a[0:3]
        s_waitcnt lgkmcnt(0)
        v_mfma_f32_16x16x32_bf16 a[0:3], v[10:13], v[24:27], a[0:3]
        s_barrier
        s_endpgm
"""
    ok = H.scan_untracked_lds_reads(planted)
    assert ok["kernels"] == 1 and ok["reads"] == 2 and ok["segments"] == 1 and ok["violations"] == []
    early = H.scan_untracked_lds_reads(planted.replace("v[20:23], v[24:27]", "v[10:13], v[24:27]"))
    assert len(early["violations"]) == 1 and "before the lgkmcnt(0)" in early["violations"][0]
    copied = H.scan_untracked_lds_reads(planted.replace("        s_waitcnt lgkmcnt(0)", "        v_mov_b32 v30, v11\n        s_waitcnt lgkmcnt(0)"))
    assert len(copied["violations"]) == 1
    drained = H.scan_untracked_lds_reads(planted.replace("        ds_read_b64_tr_b16 v[10:11], v5", "        s_waitcnt vmcnt(0)\n        ds_read_b64_tr_b16 v[10:11], v5"))
    assert len(drained["violations"]) == 1 and "vmcnt(0)" in drained["violations"][0]
    r = H.scan_library_untracked_lds_reads()
    assert r["kernels"] >= 2 and r["reads"] >= 100 and r["segments"] >= 4, r
    assert r["violations"] == [], "\n".join(r["violations"])


# scratch bytes per lane the round-3 tree is known to carry (demangled prefix -> bytes): these may shrink, never grow
_KNOWN_SCRATCH = {
    "void ctseg::conv_down_halo_kernel<ctseg::F16, 32, true, false, -1>": 28,
    "void ctseg::conv_down_halo_kernel<ctseg::BF16, 32, true, false, -1>": 24,
    "void ctseg::conv_halo_sw_kernel<ctseg::F16, 128, false, true, false>": 40,
    "void ctseg::conv_halo_sw_kernel<ctseg::BF16, 128, false, true, false>": 36,
    "void ctseg::conv_wgrad_head_kernel<4, 4>": 24,
    "void ctseg::conv_wgrad_head_kernel<4, 3>": 20,
}


def test_scratch_ratchet():
    res = H.kernel_resources()
    assert len(res) > 200
    names = list(res)
    dem = subprocess.check_output(["c++filt"], input="\n".join(names), text=True).splitlines()
    bad = []
    for sym, d in zip(names, dem):
        sc = res[sym]["scratch"]
        if sc == 0:
            continue
        allowed = max((v for k, v in _KNOWN_SCRATCH.items() if d.startswith(k)), default=0)
        if sc > allowed:
            bad.append(f"{d[:140]}: {sc} B/lane of scratch (allowed {allowed})")
    assert not bad, "kernels spilling registers to scratch (HBM traffic invisible in their own timing):\n" + "\n".join(bad)
