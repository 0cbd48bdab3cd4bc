#!/usr/bin/env python3
"""Writes tools/probes/probe_mfma_lds.hip: GEMM-shaped inner loops in hand-placed registers (one asm block per loop, so that neither
hipcc's scheduler nor its accumulator copies are part of what is measured).

Question: how much of the MFMA pipe can a weight-gradient wave tile keep busy when every bf16 operand half is a ds_read_b64_tr_b16
(the contraction runs over voxels), as a function of the wave tile (KT x CT blocks of 16x16), of software pipelining (two fragment
register sets) and of the waves per SIMD — and what do extra VALU instructions (address code) per multiply cost on top?

    python3 tools/probes/gen_probe_mfma_lds.py && hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mfma_lds.hip -o /tmp/p && /tmp/p
"""
import os

VARIANTS = [
    # name, KT, CT, read kind (None / "tr" / "b128"), schedule ("seq" / "pipe" / "mix"), VALU per multiply
    ("mfma_only_4x4", 4, 4, None, "seq", 0),
    ("tr_only_4x4", 4, 4, "tr", "reads", 0),
    ("b128_only_4x4", 4, 4, "b128", "reads", 0),
    ("tr_seq_4x4", 4, 4, "tr", "seq", 0),
    ("tr_pipe_4x4", 4, 4, "tr", "pipe", 0),
    ("tr_mix_4x4", 4, 4, "tr", "mix", 0),
    ("b128_pipe_4x4", 4, 4, "b128", "pipe", 0),
    ("b128_mix_4x4", 4, 4, "b128", "mix", 0),
    ("mfma_only_8x4", 8, 4, None, "seq", 0),
    ("tr_seq_8x4", 8, 4, "tr", "seq", 0),
    ("tr_pipe_8x4", 8, 4, "tr", "pipe", 0),
    ("tr_mix_8x4", 8, 4, "tr", "mix", 0),
    ("b128_mix_8x4", 8, 4, "b128", "mix", 0),
    ("mfma_valu1_4x4", 4, 4, None, "seq", 1),
    ("mfma_valu2_4x4", 4, 4, None, "seq", 2),
    ("mfma_valu4_4x4", 4, 4, None, "seq", 4),
    ("tr_mix_valu2_4x4", 4, 4, "tr", "mix", 2),
    ("tr_mix_valu4_4x4", 4, 4, "tr", "mix", 4),
    ("tr_mix_valu2_8x4", 8, 4, "tr", "mix", 2),
]

FRAG0 = 16          # first fragment register; set h of a KT+CT wave tile starts at FRAG0 + h * 4 * (KT + CT)
SCRATCH = 8         # v8..v15: VALU filler targets


def body(kt, ct, kind, sched, valu):
    nf = kt + ct
    lines = []

    def frag(h, i):
        return FRAG0 + h * 4 * nf + 4 * i

    def read(h, i):
        r = frag(h, i)
        if kind == "tr":
            a = "%%[a%d]" % (i & 7)
            off = (i >> 3) * 8192
            return ["ds_read_b64_tr_b16 v[%d:%d], %s offset:%d" % (r, r + 1, a, off),
                    "ds_read_b64_tr_b16 v[%d:%d], %s offset:%d" % (r + 2, r + 3, a, off + 4096)]
        if kind == "b128":
            return ["ds_read_b128 v[%d:%d], %%[lin] offset:%d" % (r, r + 3, (i & 15) * 1024)]
        return []

    def mfma(h, i, j):
        c = 4 * (i * ct + j)
        a, b = frag(h, i), frag(h, kt + j)
        out = ["v_mfma_f32_16x16x32_bf16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (c, c + 3, a, a + 3, b, b + 3, c, c + 3)]
        for v in range(valu):
            out.append("v_add_u32 v%d, v%d, v%d" % (SCRATCH + (v & 7), SCRATCH + ((v + 1) & 7), SCRATCH + ((v + 2) & 7)))
        return out

    def half(h_mul, h_read):
        out = []
        reads = [x for i in range(nf) for x in read(h_read, i)]
        muls = [mfma(h_mul, i, j) for j in range(ct) for i in range(kt)] if sched != "reads" else []
        if sched == "seq":                  # one register set: reads, wait, multiplies
            out += [x for i in range(nf) for x in read(h_mul, i)]
            out.append("s_waitcnt lgkmcnt(0)")
            for m in muls:
                out += m
        elif sched == "reads":
            out += reads
            out.append("s_waitcnt lgkmcnt(0)")
        elif sched == "pipe":               # reads of the next half first, then this half's multiplies
            out += reads
            for m in muls:
                out += m
            out.append("s_waitcnt lgkmcnt(0)")
        elif sched == "mix":                # reads of the next half spread between this half's multiplies
            per = max(1, -(-len(reads) // max(1, len(muls))))
            k = 0
            for m in muls:
                out += m
                out += reads[k:k + per]
                k += per
            out += reads[k:]
            out.append("s_waitcnt lgkmcnt(0)")
        return out

    lines += half(0, 1)
    lines += half(1, 0)
    return lines


def kernel(name, kt, ct, kind, sched, valu):
    nf = kt + ct
    nacc = 4 * kt * ct
    top = FRAG0 + 2 * 4 * nf
    pro = ["v_accvgpr_write_b32 a%d, 0" % i for i in range(nacc)]
    pro += ["v_mov_b32 v%d, 0x3f803f80" % r for r in range(SCRATCH, top)]
    loop = ["s_mov_b32 s30, %[it]", "1:"] + body(kt, ct, kind, sched, valu) + ["s_sub_u32 s30, s30, 2", "s_cmp_gt_i32 s30, 0", "s_cbranch_scc1 1b",
                                                                        "s_nop 7", "s_nop 7", "v_accvgpr_read_b32 %[r], a0"]
    text = "\\n\\t".join(pro + loop)
    clob = ", ".join(['"v%d"' % r for r in range(SCRATCH, top)] + ['"a%d"' % r for r in range(nacc)] + ['"s30"', '"scc"', '"memory"'])
    tmpl = """
__global__ __launch_bounds__(256) void k_NAME(uint32_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0x3f803f80u;
  __syncthreads();
  const int lane = threadIdx.x & 63, r16 = lane & 15, q4 = lane >> 4;
  const int mrow = 4 * q4 + (r16 >> 2), pc = (r16 & 3) * 4;
  const uint32_t base = (uint32_t)(uintptr_t)reinterpret_cast<char*>(s);
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = base + mrow * 256 + (((i ^ (mrow & 7)) << 5) + (pc << 1));
  const uint32_t lin = base + lane * 16;
  uint32_t r;
  asm volatile("TEXT"
               : [r] "=v"(r)
               : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
                 [lin] "v"(lin), [it] "s"(iters)
               : CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
"""
    return tmpl.replace("NAME", name).replace("TEXT", text).replace("CLOB", clob)


HEAD = """// GENERATED by tools/probes/gen_probe_mfma_lds.py -- do not edit.  See that file for what is measured and why.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
"""

MAIN = """
typedef void (*kern_t)(uint32_t*, int);
struct Variant { const char* name; kern_t k; int mfma, reads, valu; };
int main(int argc, char** argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  uint32_t* d; (void)hipMalloc(&d, 1 << 22);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const Variant vs[] = {%s};
  const int iters = 4000;
  printf("per half-iteration of one wave (= KT*CT multiplies + the reads of the next half); clocks at %%.0f MHz; MFMA floor = 16 clk x multiplies x waves per SIMD\\n", mhz);
  for (const Variant& v : vs) {
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, v.k, 256, 0);
    printf("%%-20s mfma %%2d reads %%2d valu %%3d  max wg/CU %%d |", v.name, v.mfma, v.reads, v.valu, occ);
    for (int w = 1; w <= 4; ++w) {
      if (w > occ) { printf("        --      "); continue; }
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(v.k, dim3(256 * w), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      const double clk = best * 1e-3 * mhz * 1e6 / iters;
      printf(" %%dw %%7.1f (%%3.0f%%%%)", w, clk, v.mfma ? 100.0 * 16.0 * v.mfma * w / clk : 0.0);
    }
    printf("\\n");
  }
  return 0;
}
"""


def main():
    out = [HEAD]
    table = []
    for name, kt, ct, kind, sched, valu in VARIANTS:
        out.append(kernel(name, kt, ct, kind, sched, valu))
        nm = kt * ct if sched != "reads" else 0
        nr = 0 if kind is None else (kt + ct) * (2 if kind == "tr" else 1)
        table.append('{"%s", k_%s, %d, %d, %d}' % (name, name, nm, nr, valu * nm))
    out.append(MAIN % ", ".join(table))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_mfma_lds.hip")
    open(path, "w").write("".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
