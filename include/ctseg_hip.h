/*
 * ctseg_hip.h — C ABI of libctseg_hip.so: the MI355X (gfx950) kernels behind the reference's
 * 3-D U-Net training step.  Plain pointers and sizes only; no torch types.  Every pointer is a
 * DEVICE pointer unless a parameter says "host".  `stream` is a hipStream_t passed as void*.
 *
 * The reference (MrinalJain17/CT-image-segmentation) has no FFI of its own: its hot path is Python
 * calling torch/MONAI ops.  Each entry point below therefore names the reference call site (or the
 * torch/MONAI op that call site executes) it replaces; INTEGRATION.md shows the ctypes binding.
 *
 * Layout convention: activations are channels-last, [N][X][Y][Z][ld] with `ld` >= C elements per
 * voxel (X,Y,Z are the reference's (H,W,D) after ToTensorV3, capstone/volumetric/transforms.py:40;
 * 2-D tensors use Z = 1).  dtype: CTSEG_F32, CTSEG_BF16 or (forward passes) CTSEG_F16 storage, fp32 accumulation always.
 *
 * Return value: 0 on success, negative on a rejected argument or a HIP launch error
 * (ctseg_last_error() gives the text).  Nothing here allocates, frees or synchronises.
 */
#ifndef CTSEG_HIP_H
#define CTSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: round 1.  2: both descriptor structs start with `struct_size` (= sizeof of the struct the CALLER was compiled
 * against); every entry point that takes a descriptor rejects a size it does not know, so a caller built against an older header
 * can never make the library read past the end of its struct (round 2 appended fields to both structs under version 1).
 * 3: ctseg_conv_desc ends with the bst_* fields (backward InstanceNorm statistics taken in the epilogue of the pass that writes
 * the gradient); ctseg_conv_bwd_stats_slots(). */
#define CTSEG_ABI_VERSION 3
#define CTSEG_F32 0
#define CTSEG_BF16 1
#define CTSEG_I16 2 /* raw-input dtypes of ctseg_resize3d_to_hwd only */
#define CTSEG_U8 3
#define CTSEG_F16 4 /* IEEE half storage, fp32 accumulation: the FORWARD (inference) passes only — ctseg_conv_igemm,
                       ctseg_instnorm_prelu_fwd, ctseg_gather_cast, ctseg_nc_to_cl / cl_to_nc, ctseg_window_gather*;
                       the backward passes and the loss gradient reject it (train in CTSEG_BF16 or CTSEG_F32) */
#define CTSEG_MAX_TAPS 27
#define CTSEG_MAX_CLASSES 8

int ctseg_abi_version(void);
const char* ctseg_last_error(void);

/* One output-parity class of an implicit-GEMM convolution pass.  A plain convolution has one class;
 * a stride-2 transposed convolution (and the input-gradient of a stride-2 convolution) has 8. */
typedef struct ctseg_conv_class {
  int32_t ntaps;                 /* K = ntaps * Cg                                            */
  int32_t kpad;                  /* padded K of this class's weight rows (multiple of 128 B)    */
  int64_t w_off;                 /* element offset of this class's [rows][kpad] weight block    */
  int32_t ox, oy, oz;            /* written voxel = row * sout + (ox,oy,oz)                     */
  int32_t taps[CTSEG_MAX_TAPS];  /* packed offsets (dx&255)|(dy&255)<<8|(dz&255)<<16, int8 each */
} ctseg_conv_class;

/* Implicit-GEMM pass: out[row-voxel][n] = bias[n] + add[..][n] + sum_{tap,c} in[row*sin+d(tap)][c] * W[n][tap*Cg+c]
 * Replaces nn.Conv3d / nn.ConvTranspose3d forward and input-gradient inside monai UNet
 * (reference capstone/models/__init__.py:3, built at capstone/volumetric/base_trainer.py:65-72). */
typedef struct ctseg_conv_desc {
  int32_t struct_size;   /* sizeof(ctseg_conv_desc) of the caller's header: checked by every entry point (ABI 2)  */
  int32_t reserved0;     /* 0                                                                                     */
  const void* in;        /* gathered tensor [N][Xi][Yi][Zi][g_ld]                                */
  const void* w;         /* packed weights (ctseg_pack_weights), rows padded to a multiple of 128 */
  const float* bias;     /* [Cn] or NULL                                                         */
  void* out;             /* written tensor [N][Xo][Yo][Zo][o_ld]                                  */
  const void* add;       /* optional addend with out's voxel indexing, [..][add_ld]; may == out    */
  float* stats;          /* optional per-tile (sum, sumsq) partials [N][stats_tiles][2][stats_ld]  */
  int32_t dtype;         /* CTSEG_F32 / CTSEG_BF16: storage of in, w (and out/add unless *_f32)    */
  int32_t N, Xi, Yi, Zi; /* gathered tensor dims                                                  */
  int32_t Xr, Yr, Zr;    /* row grid per sample                                                   */
  int32_t Xo, Yo, Zo;    /* written tensor dims                                                   */
  int32_t Cg, Cn;        /* gathered channels per tap, GEMM columns                               */
  int32_t Cn_store;      /* channels actually stored per voxel (>= Cn, pad columns are zeros)      */
  int32_t g_ld, o_ld, add_ld;
  int32_t sin, sout;
  int32_t out_f32, add_f32;
  int32_t stats_ld, stats_tiles, stats_tile0; /* partial layout; this call fills tiles [tile0, tile0+tiles*nclass) */
  int32_t nclass;
  ctseg_conv_class cls[CTSEG_MAX_CLASSES];
  /* Optional split output (where ctseg_conv_split_ok() == 1): GEMM columns >= out2_col0 are written to `out2` (same voxel
   * indexing, channel stride o2_ld, column c at channel c - out2_col0) instead of `out` — two DENSE tensors instead of
   * two interleaved channel slices of one (the fused [residual | unit0] convolution of a down ResidualUnit, the
   * [skip | sub] gradient of a SkipConnection's torch.cat).  out2 == NULL: feature off. */
  void* out2;
  int32_t out2_col0, o2_ld;
  /* Optional InstanceNorm + PReLU applied to the GATHERED operand on its way into the pass (where ctseg_conv_in_norm_ok() == 1):
   * the pass multiplies prelu((in - mean[n][c]) * rstd[n][c]) instead of in, so the normalised activation of the producing layer
   * (reference: monai Convolution's ADN "NDA" = InstanceNorm3d -> Dropout(0) -> PReLU behind capstone/models/unet.py) is never
   * written to memory; an identity-residual addend (add == in) is the transformed value as well.
   * in_mean_rstd: fp32 [N][in_norm_C][2] (mean, 1/sqrt(var + eps)) as ctseg_instnorm_finalize writes it, in_alpha: the PReLU slope
   * (one value).  NULL: feature off. */
  const float* in_mean_rstd;
  const float* in_alpha;
  int32_t in_norm_C;
  /* Optional backward statistics of the InstanceNorm + PReLU whose OUTPUT GRADIENT this pass writes (where
   * ctseg_conv_bwd_stats_slots() > 0).  The written tensor is g = dL/d prelu(xhat), xhat = (y - mean) * rstd with y the forward
   * convolution output that norm normalised (autograd of monai Convolution's ADN, reached through loss.backward() at reference
   * capstone/volumetric/base_trainer.py:80-82).  Its backward needs, per (sample, channel), sum dxhat, sum dxhat * xhat
   * (dxhat = g * prelu'(xhat)) and sum g * min(xhat, 0) (the slope gradient) — what ctseg_instnorm_prelu_bwd_reduce computes in a
   * pass of its own over (g, y).  With bst_partials set, the epilogue of THIS pass reads the matching y tile and accumulates the
   * three sums of the values it stores (after the addend, rounded to the storage type), one partial row per workgroup (or tile)
   * and sample: bst_partials[n][p][3][bst_ld], p < bst_P — exactly ctseg_instnorm_prelu_bwd_finalize's input, so the reduce
   * pass and its second read of g disappear.  Columns [bst_col0, bst_col0 + bst_C) of the pass carry channels 0..bst_C-1 of the
   * norm (bst_col0 != 0: the norm sits behind the second half of a split output, bst_col0 == out2_col0).  bst_y is indexed like
   * the written tensor (same voxels), channel stride bst_y_ld, storage dtype of the pass.  The caller zero-fills bst_partials
   * once: a workgroup only writes the rows of samples it worked on.  bst_partials == NULL: feature off. */
  const void* bst_y;
  const float* bst_mean_rstd;  /* [N][bst_C][2] as ctseg_instnorm_finalize writes it */
  const float* bst_alpha;      /* the PReLU slope (one value)                        */
  float* bst_partials;
  int32_t bst_y_ld, bst_C, bst_col0, bst_P, bst_ld;
  int32_t reserved1;
} ctseg_conv_desc;

/* Rows of the row grid / output columns one workgroup tile covers for a pass with Cn columns. */
int ctseg_conv_tile_rows(int32_t Cn);
int ctseg_conv_tile_cols(int32_t Cn);
/* InstanceNorm partial tiles per sample (summed over classes) that a pass with this geometry fills: size `stats` with it */
int ctseg_conv_num_tiles(const ctseg_conv_desc* d);
/* 1 if a pass with this geometry can take out2 / out2_col0 / o2_ld (out2_col0 must be set; pointers are ignored) */
int ctseg_conv_split_ok(const ctseg_conv_desc* d);
/* Narrow rows.  A bf16 tensor of 9..12 channels (the reference's 10 classes) may be stored 12 elements wide (24-byte rows)
 * instead of 16: the tensors either side of the logits convolution are the largest the network moves, and a quarter of
 * their bytes is padding.  g_ld == 12 with Cg == 16 (channels 12..15 read as zero), o_ld == 12 with Cn_store == 12 and
 * add_ld == 12 are accepted where this returns 1: the resident-weight 3x3x3 LDS-halo pass (any of the three) and the
 * stride-2 transposed "up" pass (output / addend only).  The InstanceNorm, loss and layout entry points take such rows as
 * they are.  The host mirror asks before it lays a tensor out this way and falls back to 16-wide rows otherwise. */
int ctseg_conv_narrow_ok(const ctseg_conv_desc* d);
/* 1 if a pass with this geometry can take in_mean_rstd / in_alpha / in_norm_C (pointers are ignored): the x-column LDS-halo pass
 * over 12-wide 16-bit rows (whose operand is staged through registers), at most 16 samples and 12 normalised channels */
int ctseg_conv_in_norm_ok(const ctseg_conv_desc* d);
/* Partial rows per sample (bst_P) the pass with this geometry fills when bst_partials is set; 0: the pass cannot take the bst_*
 * fields (run ctseg_instnorm_prelu_bwd_reduce instead).  bst_y_ld, bst_C, bst_col0 must be set; pointers other than in / out /
 * out2 / add (which select the kernel) are ignored. */
int ctseg_conv_bwd_stats_slots(const ctseg_conv_desc* d);
int ctseg_conv_igemm(const ctseg_conv_desc* d, void* stream);

/* Weight gradient: R[tap*Cg+a][b] = sum_rows in[row*sin+d(tap)][a] * dy[row][b]; row K=ntaps*Cg of R is
 * sum_rows dy[row][b] (the bias gradient).  Written as `splits` fp32 slabs [N*splits][kpad_w][cn_pad]
 * into `ws`, then ctseg_conv_wgrad_reduce sums them in fixed order (deterministic) into torch layout.
 * Replaces autograd's conv weight/bias backward for the same modules as above. */
typedef struct ctseg_wgrad_desc {
  int32_t struct_size;   /* sizeof(ctseg_wgrad_desc) of the caller's header (ABI 2) */
  int32_t reserved0;
  const void* in;        /* gathered tensor [N][Xi][Yi][Zi][g_ld]  */
  const void* dy;        /* [N][rows][d_ld], rows = Xr*Yr*Zr        */
  float* ws;             /* slabs                                    */
  int32_t dtype;
  int32_t N, Xi, Yi, Zi, Xr, Yr, Zr;
  int32_t Cg, Cn, g_ld, d_ld, sin;
  int32_t ntaps;
  int32_t taps[CTSEG_MAX_TAPS];
  int32_t splits;        /* row ranges per sample                    */
  int32_t kpad_w, cn_pad; /* slab dims: kpad_w = roundup(ntaps*Cg+1,128), cn_pad = roundup(Cn, tile cols) */
  /* optional InstanceNorm + PReLU of the gathered operand, as ctseg_conv_desc's (where ctseg_wgrad_in_norm_ok() == 1) */
  const float* in_mean_rstd;
  const float* in_alpha;
  int32_t in_norm_C;
  /* dY formed on load (ABI 3, where ctseg_wgrad_dy_norm_ok() == 1; dyn_g == NULL: off).  Columns [dyn_col0, Cn) of dY are not read
   * from `dy` (which then holds columns [0, dyn_col0) only, d_ld >= dyn_col0) but computed from the gradient dyn_g behind the
   * InstanceNorm + PReLU of those columns and its forward input dyn_y — element for element what
   * ctseg_instnorm_prelu_bwd_apply(dyn_g, dyn_y, dyn_mean_rstd, dyn_alpha, dyn_sums) would have stored, bf16 rounding included.
   * For the FIRST layer of a network (no input gradient wanted) this removes the last norm-backward pass of the step and the copy
   * of the residual gradient into a fused [d_res | d_y0] operand
   * (reference: the first ResidualUnit of the MONAI UNet built at capstone/volumetric/base_trainer.py:65-72). */
  int32_t dyn_col0;
  const void* dyn_g;           /* [N][rows][dyn_g_ld], channel c <-> column dyn_col0 + c */
  const void* dyn_y;           /* [N][rows][dyn_y_ld] */
  int32_t dyn_g_ld, dyn_y_ld;
  const float* dyn_mean_rstd;  /* [N][Cn - dyn_col0][2] */
  const float* dyn_alpha;
  const float* dyn_sums;       /* [N][Cn - dyn_col0][2]: ctseg_instnorm_prelu_bwd_finalize's output */
} ctseg_wgrad_desc;

int ctseg_wgrad_tile_cols(int32_t Cn);
/* slabs this descriptor makes ctseg_conv_wgrad write (N*splits, or one per persistent workgroup of the LDS-halo
 * kernel that few-channel 3x3x3 stride-1 bf16 layers take): size `ws` and call the reduce with it. `ws`/`dy`/`in` may be NULL here */
int ctseg_conv_wgrad_slabs(const ctseg_wgrad_desc* d);
/* Workgroups ONE slab (one sample x one row range) of this pass takes, how many of them a CU holds at a time (`per_cu`) and the
 * operand bytes one workgroup stages per 32 rows (`stage_bytes`; either may be NULL), for the caller that picks `splits`: a
 * split-K pass costs (rounds of 256 * per_cu workgroups) x (32-row stages per workgroup) x (time of a stage: its staging
 * traffic at ~24 B/clk/CU) plus the fp32 slabs it writes and the reduce reads back.  0: the pass is a persistent
 * LDS-halo kernel that sizes its own grid (splits is ignored).  `splits`, `ws`, `in`, `dy` need not be set.  (The 512-thread ring
 * kernel of the many-channel bf16 layers holds ONE 256 x 256 / 256 x 128 / 512 x 64 tile per CU; the generic kernel four
 * 128 x 128 ones.) */
int ctseg_conv_wgrad_wgs_per_slab(const ctseg_wgrad_desc* d, int32_t* per_cu, int32_t* stage_bytes);
/* 1 when this weight-gradient pass may read 12-wide bf16 rows (g_ld == 12 with Cg == 16 and / or d_ld == 12): the LDS-halo kernel */
int ctseg_wgrad_narrow_ok(const ctseg_wgrad_desc* d);
/* 1 when this weight-gradient pass can normalise its gathered operand on the fly (the 16 -> <= 16 channel LDS-halo kernel) */
int ctseg_wgrad_in_norm_ok(const ctseg_wgrad_desc* d);
/* 1 when this weight-gradient pass can form the upper columns of dY on load (dyn_*): the single-channel stride-2 first layer */
int ctseg_wgrad_dy_norm_ok(const ctseg_wgrad_desc* d);
int ctseg_conv_wgrad(const ctseg_wgrad_desc* d, void* stream);
/* dw[(b*A + a)*T + t] = sum_s ws[s][t*Astride + a][col0 + b]  for a < A, b < nb;
 * db[b] = sum_s ws[s][T*Astride][col0+b] (db may be NULL).  Astride = the pass's (padded) Cg. */
int ctseg_conv_wgrad_reduce(const float* ws, int32_t nslabs, int32_t kpad_w, int32_t cn_pad, int32_t A, int32_t Astride,
                            int32_t T, int32_t col0, int32_t nb, float* dw, float* db, void* stream);
/* The same reduction for SEVERAL weight-gradient passes in ONE launch: `jobs` is a device array of the arguments above (one entry
 * per ctseg_conv_wgrad_reduce call it replaces), results bit-identical to the per-pass calls (same partition of the slabs, same
 * fixed-order combine).  Replaces the 18 slab reduces of a training step (autograd's per-parameter gradient accumulation,
 * capstone/volumetric/base_trainer.py:80-82 -> loss.backward()) by one launch per gradient chunk: on the weight-gradient stream
 * every small launch between two GEMMs waits for the tail of the one before it.  block0 / lanes are filled by the caller:
 * lanes = 8 when nslabs >= 64 else 32 (float4 lanes along a slab row); blocks = ceil((T * Astride + 1) * ceil(nb / 4) / lanes);
 * block0 = running sum of the blocks of the jobs before it; total_blocks = the sum over all jobs.  Every job needs col0 % 4 == 0,
 * cn_pad % 4 == 0, col0 + roundup(nb, 4) <= cn_pad and a 16-byte aligned ws (ctseg_conv_wgrad_reduce_batch_ok). */
typedef struct ctseg_reduce_job {
  const float* ws;
  float* dw;
  float* db;                     /* may be NULL */
  int32_t nslabs, kpad_w, cn_pad, A, Astride, T, col0, nb;
  int32_t block0, lanes;
} ctseg_reduce_job;
int ctseg_conv_wgrad_reduce_batch_ok(const float* ws, int32_t cn_pad, int32_t col0, int32_t nb);
int ctseg_conv_wgrad_reduce_batch(const ctseg_reduce_job* jobs, int32_t n_jobs, int32_t total_blocks, void* stream);

/* dst[i] = (dtype) src[idx[i]], i < n.  With a host-built index this turns the flat fp32 parameter buffer
 * (torch Conv/ConvTranspose weight layouts) into every K-contiguous packed operand of the passes above in
 * ONE launch per optimizer step; pad entries point at a zero element of src. */
int ctseg_gather_cast(const float* src, const int32_t* idx, void* dst, int32_t dtype, int64_t n, void* stream);

/* The same re-layout, structured (ABI 3): one [rows][kpad] block of a packed operand is described instead of indexed element by
 * element.  Packed element (row n, K slot j * gs + g) of a block is the weight src[part.o + (n - part.n_lo) * part.SN +
 * (g - part.g_lo) * part.SG + tap[j]] of the part whose row / gathered-channel ranges hold (n, g), zero when no part does or
 * j >= ntaps.  A workgroup takes one row: it reads the row's source region in T-element runs (contiguous when SG == T) into LDS
 * and writes the K-contiguous row from there -- 25 MB written for ~40 MB read, where the index-driven gather reads 450 MB (every
 * gathered element sits in another cache line: consecutive K slots are T elements apart in torch's [out][in][taps] layout).
 * `rows` lists (block, row) pairs (device int32[2 * n_rows]); rows it does not list are never written (padding rows stay at the
 * zeros the caller initialised them with).  Device arrays; dtype = storage of dst (CTSEG_F32 / BF16 / F16). */
#define CTSEG_PACK_LDS_FLOATS 12288 /* sum over a block's parts of (g_hi - g_lo) * T must not exceed this (48 KB of LDS) */
typedef struct ctseg_pack_part {
  int64_t o;                     /* element offset of the weight tensor in src                          */
  int32_t n_lo, n_hi, g_lo, g_hi; /* rows / gathered channels this tensor supplies                       */
  int32_t SN, SG;                /* source strides (elements) of the row and of the gathered channel     */
} ctseg_pack_part;
typedef struct ctseg_pack_block {
  int64_t dst_off;               /* element offset of the block in dst                                   */
  int32_t kpad, gs, ntaps, T;    /* row pitch, gathered-channel stride, K slots in use = ntaps * gs, taps of the source weight */
  int32_t nparts, reserved;
  ctseg_pack_part part[2];
  int32_t tap[CTSEG_MAX_TAPS];   /* source tap id of K-slot group j                                      */
  int32_t reserved2;
} ctseg_pack_block;
int ctseg_pack_weights(const float* src, const ctseg_pack_block* blocks, int32_t n_blocks, const int32_t* rows, int32_t n_rows,
                       void* dst, int32_t dtype, void* stream);

/* InstanceNorm3d(affine=False, eps) + PReLU(1 scalar), as MONAI's Convolution block applies them
 * (SURVEY.md §3.2).  Statistics come from the conv pass's partials. */
/* partials [N][P][2][ld] fp32 -> mean_rstd [N][C][2] fp32 (biased variance, fp64 combine, fixed order) */
int ctseg_instnorm_finalize(const float* partials, int32_t N, int32_t P, int32_t ld, int32_t col0, int32_t C, double count,
                            double eps, double* scratch /* [N][64][2][ld] + N doubles; the tail zero-initialised once (completion counters) */,
                            float* mean_rstd, void* stream);
/* out = prelu((y-mean)*rstd, alpha) [+ res] ;  S = voxels per sample; mean_rstd NULL => identity norm/act skipped */
int ctseg_instnorm_prelu_fwd(int32_t dtype, const void* y, int32_t y_ld, const float* mean_rstd, const float* alpha,
                             const void* res, int32_t res_ld, void* out, int32_t out_ld, int32_t N, int64_t S, int32_t C,
                             void* stream);
/* backward, pass 1: partials [N][P][3][ld]: (sum dxhat, sum dxhat*xhat, sum g*xhat*[xhat<=0]) */
int ctseg_instnorm_prelu_bwd_reduce(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld,
                                    const float* mean_rstd, const float* alpha, float* partials, int32_t P, int32_t ld,
                                    int32_t N, int64_t S, int32_t C, void* stream);
/* partials -> sums [N][C][2] (already divided by S) and dalpha (scalar, overwritten); scratch: N*C + 1 doubles, the last
 * one a zero-initialised completion counter (the block that finishes last sums the N*C slope terms in fixed order and
 * resets it: no extra launch).  dalpha == NULL: only the per-(n,c) slope terms are left in scratch (N*C doubles suffice)
 * and ctseg_instnorm_prelu_dalpha can sum them later. */
int ctseg_instnorm_prelu_bwd_finalize(const float* partials, int32_t N, int32_t P, int32_t ld, int32_t C, double S,
                                      double* scratch, float* sums, float* dalpha, void* stream);
int ctseg_instnorm_prelu_dalpha(const double* scratch, int32_t NC, float* dalpha, void* stream);
/* backward, pass 2: dy = rstd*(dxhat - s1 - xhat*s2); optionally also copies g to g_copy (fused residual hand-off).
 * da_part != NULL: one workgroup of the launch also writes dalpha = fixed-order sum of da_part[0 .. n_da) (the per-(n,c) slope
 * terms ctseg_instnorm_prelu_bwd_finalize left in its scratch) -- the PReLU slope gradient without a launch of its own. */
int ctseg_instnorm_prelu_bwd_apply(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld,
                                   const float* mean_rstd, const float* alpha, const float* sums, void* dy, int32_t dy_ld,
                                   void* g_copy, int32_t g_copy_ld, int32_t N, int64_t S, int32_t C, const double* da_part,
                                   int32_t n_da, float* dalpha, void* stream);
/* Same pass, plus the column sums of dy over all N*S voxels -> colsum_out[C] (fp32): the bias gradient of the ConvTranspose3d
 * whose output this norm consumed (autograd: dOut.sum over voxels), without a second trip over dy.  colsum_partials: scratch of
 * P_cap rows x roundup(C, chunk) floats (P_cap >= N; a few thousand rows keep the whole chip busy); fixed-order sums. */
int ctseg_instnorm_prelu_bwd_apply_colsum(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld,
                                          const float* mean_rstd, const float* alpha, const float* sums, void* dy, int32_t dy_ld,
                                          void* g_copy, int32_t g_copy_ld, int32_t N, int64_t S, int32_t C, float* colsum_partials,
                                          int32_t P_cap, float* colsum_out, const double* da_part, int32_t n_da, float* dalpha,
                                          void* stream);

/* out[c] = sum over rows of x[row][c] (bias gradient of nn.ConvTranspose3d); partials [P][roundup(C,chunk)] fp32 scratch */
int ctseg_colsum(int32_t dtype, const void* x, int32_t ld, int64_t rows, int32_t C, float* partials, int32_t P, float* out,
                 void* stream);

/* _squash_masks_3D (capstone/volumetric/utils.py:4-7): masks u8 [B][K][S] -> labels u8 [B][S]
 * (+ optional int64 copy) and per-sample class histogram hist[B][K+1] (int64, must be zeroed by the caller). */
int ctseg_squash_masks(const uint8_t* masks, int32_t B, int32_t K, int64_t S, uint8_t* labels, int64_t* labels_i64,
                       int64_t* hist, void* stream);

/* Fused loss / metric pass over channels-last fp32 logits [B][S][ld], C classes (C <= 16).
 * Replaces F.cross_entropy (capstone/models/losses.py:53,68), softmax->argmax (capstone/training/utils.py:19-20),
 * the one-hot + compute_meandice counts (capstone/models/temp.py:173-214) and the soft-Dice / focal sums of
 * monai DiceLoss / FocalLoss.  do_stats: per-workgroup partials part[B][P][2+3C] doubles
 *   (sum w*nll, sum w, then per class sum p, sum p*y, sum focal) and integer counts cnt[B][3][C]
 *   (|pred==c & true==c|, |pred==c|, |true==c|; int64, zeroed by the caller).
 * do_stats / do_grad = 2 select the cross-entropy-only fast path (soft sums left 0, gradient = ce term only).
 * do_grad: dlogits = coef-weighted gradient, coef[B][1+3C] fp32 = (ce_scale, a[C], b[C], f[C]):
 *   ce_scale*w[t]*(p-onehot) + softmax-jacobian of (a_c*y_c + b_c) + focal term f_t.  dlogits dtype = gdtype. */
int ctseg_seg_loss(const float* logits, int32_t ld, const uint8_t* labels, int32_t B, int64_t S, int32_t C,
                   const float* class_weight, int32_t do_stats, double* part, int32_t P, int64_t* cnt, int32_t do_grad,
                   const float* coef, void* dlogits, int32_t g_ld, int32_t gdtype, uint8_t* pred_out, void* stream);
/* The logits convolution of the U-Net's head with the cross-entropy of the training step fused into its epilogue: the fp32 logits
 * (1.2 GB per step at 2 x 512 x 512 x 48) are never written or re-read.  Replaces, for the native training step, the pair
 *   ctseg_conv_igemm(d) [monai UNet model.2.1.conv.unit0 + identity residual]  +  ctseg_seg_loss(do_stats = do_grad = 2)
 *   [F.cross_entropy, capstone/models/losses.py:45-68; softmax -> argmax + Dice counts, training/utils.py:19-20, models/metrics.py:15-21]
 * with the same per-voxel arithmetic: dlogits and cnt are bit-identical to the two-call path, the loss sums differ in summation
 * order only.  d: the recorded descriptor of that convolution (d->out is not written).  labels [N][S] u8; class_weight [C] or NULL;
 * coef[n * coef_stride] = d(loss)/d(weighted NLL sum of sample n); dlogits [N][S][g_ld] in d->dtype; part [N][P][R] doubles, this
 * launch writes slots 0 .. ctseg_conv_logits_ce_slots()-1 of every sample (entries 0 = weighted NLL sum, 1 = weight sum, rest 0);
 * cnt [N][3][C] += counts (zeroed by the caller).  ctseg_conv_logits_ce_slots: 0 when the pass is not eligible (the caller then
 * keeps the two-call path), else the number of partial slots per sample the launch fills. */
int ctseg_conv_logits_ce_slots(const ctseg_conv_desc* d, int32_t C);
int ctseg_conv_logits_ce(const ctseg_conv_desc* d, const uint8_t* labels, int32_t C, const float* class_weight, const float* coef,
                         int32_t coef_stride, void* dlogits, int32_t g_ld, double* part, int32_t P, int32_t R, int64_t* cnt,
                         void* stream);
/* cnt[B][3][C] += (|pred==c & true==c|, |pred==c|, |true==c|) from two u8 label maps [B][S]
 * (DiceMetricWrapper on squashed predictions, capstone/models/metrics.py:15-31); cnt zeroed by the caller. */
int ctseg_dice_counts(const uint8_t* pred, const uint8_t* truth, int32_t B, int64_t S, int32_t C, int64_t* cnt, void* stream);
/* part [B][P][R] doubles -> out [B][R] doubles, fixed order */
int ctseg_reduce_partials_f64(const double* part, int32_t B, int32_t P, int32_t R, double* out, void* stream);
/* One launch for the scalars a cross-entropy training step logs (capstone/volumetric/base_trainer.py:80-82, 98-110 ->
 * capstone/models/losses.py:45-68 and metrics.py:15-31): out[0] = sum_b red[b][0] / sum_b red[b][1] with red = the reduced
 * ctseg_seg_loss records (entry 0 = weighted NLL sum, 1 = weight sum); out[2 + k] = Dice of class k+1 averaged over the
 * samples whose truth holds it (0 if none), from cnt[B][3][C]; out[1] = mean of those C-1 values.  out: 1 + C floats. */
int ctseg_loss_dice_summary(const double* red, int32_t B, int32_t R, const int64_t* cnt, int32_t C, float* out, void* stream);

/* torch.optim.Adam step (capstone/volumetric/base_trainer.py:113-114) on flat fp32 buffers.  lr / betas / eps are doubles (ABI 3),
 * as the Python floats torch receives: 1 - beta2 is formed in double and THEN rounded (0.001f; 1.f - 0.999f is 1.3e-5 off). */
int ctseg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                    int32_t step, float grad_scale, void* stream);

/* x[0..n) *= host_scale * (dev_scale ? *dev_scale : 1), in place; a launch whose factor is exactly 1 touches no memory.
 * Two users: the upstream gradient autograd hands ``loss.backward()`` (capstone/volumetric/base_trainer.py:80-82 returns the
 * loss; Lightning / AMP may scale it) applied to the d loss / d logits the fused loss pass wrote for an upstream gradient of 1,
 * and the 1 / world of the data-parallel gradient MEAN applied to the all-reduced flat gradient on the drop-in path.
 * dtype: CTSEG_F32 or CTSEG_BF16. */
int ctseg_scale_inplace(void* x, int32_t dtype, int64_t n, const float* dev_scale, float host_scale, void* stream);

/* Layout / dtype plumbing. */
int ctseg_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream);
/* fp32 [N][C][S] (torch NC*) -> dtype [N][S][ld] channels-last (pad channels zeroed) and back (fp32 out) */
int ctseg_nc_to_cl(const float* src, void* dst, int32_t dtype, int32_t N, int32_t C, int64_t S, int32_t ld, void* stream);
int ctseg_cl_to_nc(const void* src, int32_t dtype, float* dst, int32_t N, int32_t C, int64_t S, int32_t ld, void* stream);

/* 3-D input pipeline on the device: one pass per instance replaces
 *   Resize3D.apply / apply_to_mask   capstone/volumetric/transforms.py:14-22  (F.interpolate, mode "nearest":
 *                                    src = min((int)floorf(dst * ((float)in / out)), in - 1) per axis)
 *   ToTensorV3.apply / apply_to_mask capstone/volumetric/transforms.py:39-43  ((C,D,H,W) -> (C,H,W,D))
 *   optionally _squash_masks_3D      capstone/volumetric/utils.py:4-7         (labels_out, + per-class voxel counts in hist[K+1],
 *                                    which the caller zeroes) and the HU window of capstone/transforms/transforms_2d.py:97-107
 *                                    (window 0: none, 1: clip to [lo,hi], 2: clip and map to [0,1]; the 3-D reference path has none).
 * image [D][H][W] of image_dtype (CTSEG_F32 / CTSEG_I16 / CTSEG_U8) -> image_out fp32 [Ho][Wo][Do];
 * masks [K][D][H][W] u8 -> masks_out [K][Ho][Wo][Do] and/or labels_out [Ho][Wo][Do].  image or masks may be NULL. */
int ctseg_resize3d_to_hwd(const void* image, int32_t image_dtype, const uint8_t* masks, int32_t K, int32_t D, int32_t H, int32_t W,
                          int32_t Do, int32_t Ho, int32_t Wo, int32_t window, float win_lo, float win_hi, float* image_out,
                          uint8_t* masks_out, uint8_t* labels_out, int64_t* hist, void* stream);

/* Sliding-window inference (BASELINE.json configs[4]).  The reference has no inferer (grep: 0 hits); the
 * semantics are those of MONAI 0.3 `monai.inferers.sliding_window_inference`, the companion of the `monai.networks.nets.UNet`
 * the reference builds at capstone/volumetric/base_trainer.py:65-72.
 *  gather: vol fp32 [Cin][X][Y][Z] -> dst dtype [rx][ry][rz][ld] = the window at (x0,y0,z0); voxels outside the volume = cval
 *          (MONAI pads volumes smaller than the ROI; starts may be negative), pad channels = 0.
 *  blend : out[x0+x][y0+y][z0+z][c] += importance[x][y][z] * inv_count[dst voxel] * logits[x][y][z][c] for voxels inside
 *          the volume (inv_count == NULL: factor 1 — used to accumulate the weight sum itself); one window per launch, so
 *          overlapping windows accumulate in stream order (deterministic). */
int ctseg_window_gather(const float* vol, int32_t Cin, int32_t X, int32_t Y, int32_t Z, int32_t x0, int32_t y0, int32_t z0,
                        int32_t rx, int32_t ry, int32_t rz, float cval, void* dst, int32_t dtype, int32_t ld, void* stream);
/* The same for nw windows in one launch: starts[nw][3] (device) = window origins, dst = [nw][rx*ry*rz][ld]. */
int ctseg_window_gather_batch(const float* vol, int32_t Cin, int32_t X, int32_t Y, int32_t Z, const int32_t* starts, int32_t nw,
                              int32_t rx, int32_t ry, int32_t rz, float cval, void* dst, int32_t dtype, int32_t ld, void* stream);
int ctseg_window_blend(const float* logits, int32_t ld, int32_t C, int32_t rx, int32_t ry, int32_t rz, int32_t x0, int32_t y0,
                       int32_t z0, const float* importance, const float* inv_count, float* out, int32_t X, int32_t Y, int32_t Z,
                       int32_t out_ld, void* stream);
/* The same accumulation for ALL nw windows of one forward batch in one launch (logits [nw][rx*ry*rz][ld], starts[nw][3] =
 * window origins in output coordinates, on the device): per output voxel the covering windows are added in index order, i.e.
 * exactly the sums of nw ctseg_window_blend calls in that order.  Rows are 16-byte vectors: ld == out_ld, a multiple of 4, <= 16.
 * bbox (HOST pointer, 6 ints: origin x,y,z and extents, inside the volume; NULL = whole volume) bounds the voxels swept: it must
 * contain every window of the batch clipped to the volume. */
int ctseg_window_blend_batch(const float* logits, int32_t ld, int32_t C, int32_t rx, int32_t ry, int32_t rz, const int32_t* starts,
                             int32_t nw, const float* importance, const float* inv_count, float* out, int32_t X, int32_t Y,
                             int32_t Z, int32_t out_ld, const int32_t* bbox, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTSEG_HIP_H */
