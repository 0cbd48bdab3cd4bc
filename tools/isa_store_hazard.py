#!/usr/bin/env python3
"""Scan the gfx950 code objects INSIDE lib/libctseg_hip.so (what ships, not a recompilation) for the wide-store hazard of
DESIGN.md section 3.2g "Hardware fact 1":

    buffer_store_dwordx3 / buffer_store_dwordx4 with an SGPR soffset, followed within two issue slots by a VALU instruction
    that writes one of the store's DATA registers, without an s_nop in between.

LLVM's hazard recogniser pads this case only for stores WITHOUT an SGPR offset (GCNHazardRecognizer: "this hazard only exists if
the instruction is not using a register in the soffset field"); on gfx950 a handful of 16-bit elements per launch came out as
garbage with one.  The two stores that hit it keep their data registers live across `s_nop 1`; nothing else guards the next
compiler upgrade or the next kernel — this scanner does (tests/test_isa_hazards.py runs it on the built library).

    python tools/isa_store_hazard.py [path/to/libctseg_hip.so]        # prints a per-bundle summary; exit code 1 on a hazard
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "ct-image-segmentation_amd", "lib", "libctseg_hip.so")

_REG = re.compile(r"^([va])(?:(\d+)|\[(\d+):(\d+)\])$")


def _regset(op):
    """'v5' / 'v[10:13]' / 'a[0:3]' -> set of (file, index); anything else -> empty"""
    m = _REG.match(op.strip())
    if not m:
        return set()
    f = m.group(1)
    if m.group(2) is not None:
        return {(f, int(m.group(2)))}
    return {(f, i) for i in range(int(m.group(3)), int(m.group(4)) + 1)}


def code_objects(lib_path, arch="gfx950"):
    """yield (bundle index, bytes) of every `arch` code object in the library's .hip_fatbin section (one offload bundle per
    translation unit with -fno-gpu-rdc)"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        blob = open(fat, "rb").read()
    for bi, m in enumerate(re.finditer(re.escape(MAGIC), blob)):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if arch in triple and size > 0:
                yield bi, blob[base + off:base + off + size]


def disassemble(obj_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(obj_bytes)
        f.flush()
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], text=True)


def _instructions(text):
    """[(kernel symbol, mnemonic, [operands])] in program order"""
    out, sym = [], "?"
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln.strip())
        if m:
            sym = m.group(1)
            continue
        ln = ln.split("//")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("Disassembly") or "file format" in ln:
            continue
        parts = ln.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        out.append((sym, mn, ops))
    return out


def scan_text(text, slots=2):
    """-> (number of wide SGPR-offset stores seen, [hazard descriptions])"""
    ins = _instructions(text)
    wide, hazards = 0, []
    for i, (sym, mn, ops) in enumerate(ins):
        if not re.match(r"^buffer_store_(dwordx[34]|b96|b128)$", mn) or len(ops) < 4:
            continue
        soff = ops[3].split()[0]
        if not re.match(r"^(s\d+|m0|ttmp\d+)$", soff):
            continue                       # immediate 0 / literal offset: the case LLVM pads by itself
        wide += 1
        data = _regset(ops[0])
        for k in range(1, slots + 1):
            if i + k >= len(ins):
                break
            s2, mn2, ops2 = ins[i + k]
            if mn2 == "s_nop":
                break                      # wait states inserted: what follows is safe
            if s2 != sym:
                break
            if mn2.startswith("v_") and ops2 and (_regset(ops2[0]) & data):
                hazards.append(f"{sym}: `{mn} {', '.join(ops)}` then (+{k}) `{mn2} {', '.join(ops2)}` writes its data registers")
                break
    return wide, hazards


def scan_library(lib_path=DEFAULT_LIB):
    """-> {"bundles": n, "kernels_text_bytes": ..., "wide_sgpr_stores": n, "hazards": [...]}"""
    res = {"bundles": 0, "wide_sgpr_stores": 0, "instructions": 0, "hazards": []}
    for bi, obj in code_objects(lib_path):
        text = disassemble(obj)
        w, hz = scan_text(text)
        res["bundles"] += 1
        res["wide_sgpr_stores"] += w
        res["instructions"] += len(_instructions(text))
        res["hazards"] += [f"bundle {bi}: {h}" for h in hz]
    return res


def disassemble_with_labels(obj_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(obj_bytes)
        f.flush()
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", f.name], text=True)


def _blocks(text, kernel_substr):
    """{kernel symbol: [block]} with block = {"label": str | None, "ins": [(mnemonic, [operands])], "succ": [block indices]};
    text from disassemble_with_labels() (branch targets are `<L12>:` lines / `L12` operands)"""
    kernels, cur, sym = {}, None, None
    for ln in text.splitlines():
        st = ln.strip()
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", st)
        if m:
            name = m.group(1)
            if re.match(r"^L\d+$", name):
                if sym is not None:
                    cur = {"label": name, "ins": [], "succ": []}
                    kernels[sym].append(cur)
                continue
            sym = name if kernel_substr in name else None
            if sym is not None:
                cur = {"label": None, "ins": [], "succ": []}
                kernels[sym] = [cur]
            continue
        if sym is None:
            continue
        st = st.split("//")[0].strip()
        if not st or st.startswith("Disassembly") or "file format" in st:
            continue
        parts = st.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur["ins"].append((mn, ops))
        if mn.startswith("s_cbranch") or mn in ("s_branch", "s_endpgm"):
            cur = {"label": None, "ins": [], "succ": []}
            kernels[sym].append(cur)
    for sym, bl in kernels.items():
        by_label = {b["label"]: i for i, b in enumerate(bl) if b["label"]}
        for i, b in enumerate(bl):
            last = b["ins"][-1] if b["ins"] else ("", [])
            if last[0] != "s_endpgm" and last[0] != "s_branch" and i + 1 < len(bl):
                b["succ"].append(i + 1)
            if (last[0].startswith("s_cbranch") or last[0] == "s_branch") and last[1] and last[1][-1] in by_label:
                b["succ"].append(by_label[last[1][-1]])
    return kernels


def _sources(mn, ops):
    if mn.startswith("ds_write") or mn.startswith("buffer_store") or mn.startswith("global_store") or mn.startswith("v_cmp") or mn.startswith("s_"):
        return ops
    return ops[1:]


def scan_untracked_lds_reads(text, kernel_substr="conv_wgrad_ring_kernel"):
    """Kernels that read LDS through inline assembly (ds_read_b64_tr_b16 in conv_wgrad_ring.hip: the compiler neither orders those
    reads against the direct-to-LDS loads nor waits for their data) must do by hand what the compiler does for its own loads.
    Per kernel whose symbol contains ``kernel_substr``, over its control-flow graph (text from disassemble_with_labels()):
      * a register written by such a read may not be READ (by a multiply, a move, anything) on any path before an `s_waitcnt`
        with lgkmcnt(0) — forward dataflow of the set of registers with a read in flight, union over predecessors;
      * a straight-line stretch between two barriers that holds both transposed reads and direct-to-LDS loads has no
        `s_waitcnt vmcnt(0)` (that wait would empty the ring every step — the reason the reads are inline assembly at all).
    -> {"kernels": n, "reads": n, "segments": n, "violations": [...]}"""
    res = {"kernels": 0, "reads": 0, "segments": 0, "violations": []}
    for sym, bl in _blocks(text, kernel_substr).items():
        res["kernels"] += 1
        entry = [set() for _ in bl]

        def walk(i, report):
            pending = set(entry[i])
            for mn, ops in bl[i]["ins"]:
                if mn == "s_waitcnt":
                    if "lgkmcnt(0)" in " ".join(ops):
                        pending = set()
                    continue
                for o in _sources(mn, ops):
                    hit = _regset(o.split()[0] if o else o) & pending
                    if hit and report:
                        res["violations"].append(f"{sym}: `{mn} {', '.join(ops)}` reads {sorted(hit)[:2]} before the lgkmcnt(0) that covers its ds_read_b64_tr_b16")
                if mn in ("ds_read_b64_tr_b16", "ds_read_b64"):          # (every LDS read of these kernels is inline assembly)
                    pending |= _regset(ops[0])
                    if report:
                        res["reads"] += 1
                elif ops and not mn.startswith("ds_write") and not mn.startswith("buffer_store") and not mn.startswith("global_store"):
                    pending -= _regset(ops[0].split()[0])          # (overwritten by something the compiler tracks)
            return pending

        changed = True
        while changed:
            changed = False
            for i in range(len(bl)):
                out = walk(i, False)
                for j in bl[i]["succ"]:
                    if not out <= entry[j]:
                        entry[j] |= out
                        changed = True
        for i in range(len(bl)):
            walk(i, True)
        # ring steps: the stretch from a barrier to the next one (or to the end of its basic block: a step may end in a branch)
        for b in bl:
            reads = dma = drain = 0

            def close():
                if reads and dma:
                    res["segments"] += 1
                    if drain:
                        res["violations"].append(f"{sym}: s_waitcnt vmcnt(0) in a ring step that both reads fragments and requests a stage")

            for mn, ops in b["ins"]:
                if mn == "s_barrier":
                    close()
                    reads = dma = drain = 0
                elif mn == "s_waitcnt" and "vmcnt(0)" in " ".join(ops):
                    drain += 1
                elif mn == "ds_read_b64_tr_b16":
                    reads += 1
                elif mn.startswith("buffer_load") and any("lds" in o for o in ops):
                    dma += 1
            close()
    return res


def scan_library_untracked_lds_reads(lib_path=DEFAULT_LIB):
    tot = {"kernels": 0, "reads": 0, "segments": 0, "violations": []}
    for bi, obj in code_objects(lib_path):
        r = scan_untracked_lds_reads(disassemble_with_labels(obj))
        for k in ("kernels", "reads", "segments"):
            tot[k] += r[k]
        tot["violations"] += r["violations"]
    return tot


def kernel_resources(lib_path=DEFAULT_LIB):
    """{demangled-ish kernel symbol: {"scratch": bytes per lane, "vgpr": n, "spills": n, "lds": bytes}} from the code objects'
    AMDGPU metadata notes — DESIGN.md section 3.2f: a 112 B/lane spill was invisible in the kernel's own timing and cost +0.9 GB of
    HBM traffic per launch, so ScratchSize is part of the check list"""
    out = {}
    for bi, obj in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], text=True)
        for blk in notes.split("- .agpr_count")[1:]:
            g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
            out[g("name")] = {"scratch": int(g("private_segment_fixed_size")), "vgpr": int(g("vgpr_count")),
                              "spills": int(g("vgpr_spill_count")), "lds": int(g("group_segment_fixed_size"))}
    return out


if __name__ == "__main__":
    r = scan_library(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB)
    print(f"{r['bundles']} gfx950 code objects, {r['instructions']} instructions, {r['wide_sgpr_stores']} wide buffer stores with an SGPR "
          f"soffset, {len(r['hazards'])} hazards")
    for h in r["hazards"]:
        print("  HAZARD", h)
    sys.exit(1 if r["hazards"] else 0)
