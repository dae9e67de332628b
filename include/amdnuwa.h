/*
 * libamdnuwa -- C-ABI of the MI355X (gfx950) NUWA video-decoder training hot path.
 *
 * The reference (lucidrains/nuwa-pytorch) has no FFI / operator registry: its boundary for this
 * path is the set of nn.Module classes in nuwa_pytorch/nuwa_pytorch.py (np.py) and
 * nuwa_pytorch/vqgan_vae.py (vq.py).  Each entry point below replaces the *body* of one of
 * those modules' forward (or its autograd backward) and cites the lines it replaces.  The
 * Python package `nuwa_pytorch_amd` keeps the module classes / signatures / state-dict keys
 * and calls these functions through ctypes (see INTEGRATION.md for the reference-side stub).
 *
 * Conventions
 *   - plain pointers to DEVICE memory + sizes + an explicit hipStream_t; nothing allocates,
 *     nothing synchronises, nothing throws; return 0 on success, <0 = AMDNUWA_ERR_*, >0 = hipError_t.
 *   - bf16 tensors are passed as uint16_t*; "hi/lo" pairs are the bf16 split of an fp32 value
 *     (value ~= hi + lo).  A NULL lo pointer selects the plain-bf16 operand path; non-NULL lo
 *     on both operands selects the 3-MFMA parity path (hi*hi + hi*lo + lo*hi).
 *   - activations are row-major [rows, features] with an explicit leading dimension in elements.
 *   - token rows of the video decoder: row = sample * ntok + i, i = 0 is <bos>, i = 1 + p is the
 *     token at raster position p = (f*H + y)*W + w.
 */
#ifndef AMDNUWA_H
#define AMDNUWA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* amdnuwa_stream;   /* == hipStream_t */

int amdnuwa_abi_version(void);                 /* bumps when any signature or documented argument meaning below changes (13: tuning keys 0..31,
                                                * amdnuwa_xattn_unpack's flag bit 1, chunk-permuted dS / Pm columns of amdnuwa_xattn2_bwd;
                                                * 14: amdnuwa_linear_ce_x3 added; 15: the two-MFMA products -- amdnuwa_gemm_desc.ab_f16 with Blo, amdnuwa_gemm_nt_f16x2_supported,
                                                *     o_lo_f16 on the two fp16 forward cores; 16: the fp16-gradient backward; 17: amdnuwa_gemm_desc.a_chunk32, amdnuwa_gemm_tn_chunked_a_supported,
                                                *     amdnuwa_xattn2_bwd_ex, AMDNUWA_LN_RESID_MINUS, tuning key 25; 18: the amdnuwa_xattn6_* family; 19: amdnuwa_sparse3dna_bwd_f16, amdnuwa_xattn6_bwd_f16, amdnuwa_xattn6_pack_bwd_f16, o == NULL in the two fp16 forward cores,
                                                *     ab_f16 on the whole-M TN kernel with alpha_dev in its direct epilogue, c_f16 on the two-MFMA NT product) */
const char* amdnuwa_error_string(int code);
/* runtime tuning knobs (A/B benchmarking only; 0 = the library's auto policy everywhere):
 *   key 0  NT GEMM variant: 2 direct-to-LDS BK 32 (128x128 tiles), 5 register-staged 128x128, 7 the 256x256 ring family for every size,
 *          11 its K-step 64 form with staggered wave rows, 12 the four-wave long-K kernel for every K % 64 == 0
 *          (1, 3, 4, 6, 8, 9, 10: A/B variants of rounds 2-4, removed in round 6 -- 1 runs as 2, 10 as 0, the others as 7)
 *   key 1  TN split-K workgroup slots        key 2  TN minimum token rows per split
 *   key 3  Sparse3DNA forward: 1 = keep the VALU (dot2) kernel        key 13 hi + lo NT GEMM: 1 = first-generation 128x128 kernel
 *   key 4  Sparse3DNA backward: 1 = VALU kernels, 2 = MFMA query side + VALU key side, 4 = MFMA query side + RECOMPUTING MFMA key side
 *          (no ds / P' workspace: 1.5x instead of 2.85x the algorithmic HBM bytes, 15-22 % slower; 0 = MFMA kernels with the workspace)
 *   key 5  cross-attention forward: 1 = generic (not unrolled) kernel
 *   key 6  TN GEMM variant: 1 register-staged, 2 direct-to-LDS 128x128, 3 256x256 ring
 *   key 7  NT probe: bit 0 skips the epilogue stores, bit 1 skips the main loop (tools/gemm_probe.py; results are garbage)
 *   key 8  TN 256x256 ring: 2 = staggered wave rows instead of the lock-step schedule
 *   key 9  3DNA MFMA forward probe: bits 0 / 1 / 2 skip the score / softmax+mix / apply phase (garbage results), bit 3 = fragment-shaped
 *          key loads in the score pass instead of the staged ones
 *   key 14 NT start-phase step in ~0.25 us units (0 = off)        key 15 VAE kernels: 1 = first (VALU) forms
 *   key 16 Sparse3DNA MFMA forward: query rows per workgroup (0 = auto = 2 where H splits into dh * rows; 1 / 2 / 4 forced): a tile of rows
 *          of one residue class of y stages every key / value row once for the rows that tap it
 *   key 17 3DNA backward timing probes (garbage results): bit 0 no score sweeps, 1 no ds / P' workspace stores, 2 no dq apply sweep,
 *          3 no dW_th sums; key side: bit 4 no coefficient gathers, 5 no q / dO row fetch
 *   key 18 (unused since ABI 18: the cross-attention timing probes were run-time branches inside the chunk loops -- removed)
 *   key 19 3DNA MFMA query-side backward: 1 = the three separate item passes (P' mix, dW_th, dP mix) instead of the fused one
 *   key 20 NT 256x256 ring as a PERSISTENT kernel (one workgroup per CU walks the tile list, the DMA ring runs on across tile borders):
 *          0 = auto (more tiles than CUs and K <= 1024), 1 = never, 2 = always
 *   key 22 NT long-K kernel (four waves of 128x128, K-step 64, hand-placed main loop: gemm_nt_w4k_kernel): 0 = auto (K >= 1024 and K % 64 == 0),
 *          1 = never (8-wave ring), 2 = for every K % 64 == 0        key 23 TN four-wave kernel (gemm_tn_w4k_kernel): 1 = keep the 8-wave ring
 *   key 24 Sparse3DNA MFMA backward workspace: 0 = ONE array of (bf16 ds | bf16 P') words (round 5), 1 = the two fp32 arrays (same dK / dV bits)
 *   key 25 batched narrow TN (N <= 64, 128 < M <= 384: the cross attention's dK / dV): 1 = 128-row tiles instead of one workgroup per batch element,
 *          2 = always through the split-K reduction (no direct store of a one-split result), 3 = the general whole-M kernel also where the lean
 *          form applies (whole 32-row K-steps per split, plane offsets below 4 GiB: gemm_tn_wmf_kernel, bit-identical results)
 * (keys run 0..31; the K-step 64 form of the 256x256 ring stages full 128-byte DMA lines in two 64 KiB stages, `auto` uses it for
 *  1024 <= K < 2048; any non-zero key 0 disables the few-row weight-streaming path that M <= 32 normally takes) */
int amdnuwa_set_tuning(int key, int value);
/* fp16 saturation monitor: number of threads (since the last reset) that handed a value beyond +-65504 to one of the library's saturating
 * COUNTED fp16 stores: the LayerNorm fp16 copies, the fp16-gradient epilogues (GEMM EPI 4 / 5, LayerNorm backward) and the fp16 copy of
 * the cross-attention output (amdnuwa_xattn6_fwd).  The forward GEMM epilogue copies (q / k / v, the FeedForward gate), the Sparse3DNA output
 * copy and amdnuwa_xattn2_fwd_f16 clamp to +-65504 as well but are NOT counted (one more live register there sends the persistent ring to
 * scratch): their inputs are LayerNorm outputs times weights inside the checked fp16 range.  0 in a healthy run: a saturated value is DEFINED
 * (clamped) but no longer the reference's arithmetic.  Synchronises the device; reset != 0 clears the counters. */
unsigned long long amdnuwa_f16_sat_count(int reset);
int amdnuwa_get_tuning(int key);

/* opt-in HIP-event launch timer: while armed, _begin/_end bracket one launch on `stream` with an
 * event pair; _collect synchronises and returns the summed kernel time and the launch count. */
void amdnuwa_timer_arm(int on);
int amdnuwa_timer_begin(amdnuwa_stream stream);
int amdnuwa_timer_end(amdnuwa_stream stream);
int amdnuwa_timer_collect(double* total_ms, long long* launches);
/* ABI 18: the same per launch -- ms[i] = duration of the i-th bracketed launch since the last collect (cap >= their number) */
int amdnuwa_timer_collect_each(double* ms, long long cap, long long* launches);

/* ------------------------------------------------------------------------------------------
 * GEMM (replaces every nn.Linear on the path: np.py:274-277 FeedForward, np.py:311-313 Attention,
 * np.py:401-405 Sparse3DNA, np.py:1819 to_logits; and their autograd backward)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint16_t* A; const uint16_t* Alo; long long strideA; int lda;
    const uint16_t* B; const uint16_t* Blo; long long strideB; int ldb;
    void* C; uint16_t* Clo; long long strideC; int ldc;
    int c_is_bf16;          /* 0: C is float*, 1: C (and optional Clo) are bf16 */
    const float* bias;      /* NT only, fp32 output only; may be NULL */
    float alpha, beta;      /* beta: TN only (C = beta*C + alpha*A^T B) */
    int M, N, K;
    int batch;              /* >=1; operand/result strides are per batch element */
    int shift_ntok;         /* >0: fold ShiftVideoTokens (np.py:185-253) into the loader of the */
    int shift_fmap;         /*     activation operand (A for NT, B for TN); ntok = rows per sample */
    int batch_inner;        /* >0: two-level batch: element z lives at (z / batch_inner) * stride +   */
    long long strideA_inner, strideB_inner, strideC_inner;   /*  (z % batch_inner) * stride_inner      */
    /* NT, bf16 C, batch <= 1: GEGLU gate (np.py:255-258) applied to the product.  C = u [M, N] is in the interleaved-by-8 layout
     * (see amdnuwa_geglu_il_fwd) and C2 [M, N/2] (bf16 hi[/lo], row pitch ldc2) = a * gelu_erf(gate); NULL = plain GEMM. */
    uint16_t* C2; uint16_t* C2lo; int ldc2;
    /* ... and its backward: with geglu_u != NULL the product is dgg = d(a * gelu(gate)) [M, N] and C2 [M, 2N] (pitch ldc2)
     * receives du = (dgg * gelu(gate) | dgg * a * gelu'(gate)) in the interleaved layout, computed from geglu_u [M, 2N]
     * (pitch ld_u, hi[/lo]).  When amdnuwa_gemm_nt_fused(d) != 0 this happens in the GEMM epilogue and C is NOT written;
     * otherwise C receives dgg and the stand-alone kernel follows. */
    const uint16_t* geglu_u; const uint16_t* geglu_u_lo; int ld_u;
    /* NT, bf16 C, hi + lo operands: with c_lo_f16 != 0 Clo receives the fp16 rendering of the FULL product value (not the bf16
     * residual) -- the operand form of the fp16 attention cores of the 'bf16x3-fwd' mode (amdnuwa_sparse3dna_fwd_f16,
     * amdnuwa_xattn2_fwd_f16): q / k / v leave the projection GEMM as a bf16 copy (for the bf16 backward) and an fp16 copy (for the
     * forward core).  Only the 256x256 hi + lo ring does this: ask amdnuwa_gemm_nt_f16_fused() first; otherwise run the plain
     * hi + lo product and amdnuwa_hilo_to_f16(). */
    int c_lo_f16;
    /* NT: ab_f16 != 0: A and B hold FP16 values (Alo / Blo / Clo must be NULL) and the product runs on the fp16 MFMA.  C: fp32, or
     * bf16 with the optional GEGLU output computed on the fp32 accumulators: C2 receives its FP16 copy (the next GEMM's A operand),
     * C2lo -- if not NULL -- its bf16 copy (the backward's operand).  Without C2, a non-NULL Clo receives the FP16 copy of the bf16
     * output itself (q / k / v: bf16 for the backward, fp16 for the forward attention core).  The FeedForward GEMMs (reference nuwa_pytorch.py:255-286) of
     * the 'bf16x3-fwd' forward.  256x256 ring only: amdnuwa_gemm_nt_f16ops_supported() first, AMDNUWA_ERR_UNSUPPORTED otherwise. */
    int ab_f16;
    /* ab_f16 with Blo != NULL: the TWO-MFMA form -- A = fp16 values (an activation rounded to 11 significand bits), B / Blo = an fp16
     * hi + lo pair with hi + lo = the fp32 weight to ~22 bits; per 32-chunk of k the products lo*a, hi*a accumulate in fp32.  C: fp32
     * (+ bias), or bf16 whose optional Clo receives the FP16 rendering of the output.  What the 'bf16x3-fwd' mode runs for to_out, the
     * cross-attention q / kv projections and to_logits (reference nuwa_pytorch.py:370-379, 611-613, 1956) when its two-MFMA switch is
     * on.  amdnuwa_gemm_nt_f16x2_supported() first; AMDNUWA_ERR_UNSUPPORTED otherwise. */
    /* Round 5 (ABI 16), the fp16-gradient backward.  NT with ab_f16 and c_is_bf16: c_f16 != 0 -> C (and the GEGLU-backward output C2) hold
     * FP16 values, saturating, instead of bf16 (Clo / C2lo NULL): dgrad products whose operands are fp16(S * gradient) and an fp16 weight.
     * The GEGLU backward (geglu_u) is available on the fp16-operand ring as well then (geglu_u stays bf16: it is read element-wise).
     * TN: ab_f16 != 0 -> A and B hold fp16 values (weight gradients from fp16 gradients and the fp16 activation copies).
     * alpha_dev (TN): optional DEVICE scalar multiplied into alpha by the split-K reduction (1 / S of the gradient scale). */
    int c_f16;
    const float* alpha_dev;
    /* ABI 17, TN only: a_chunk32 != 0 -> A is stored in planes of 32 columns, element (token row g, column c) at
     * A[(c / 32) * K * 32 + g * 32 + c % 32] (lda unused; batch strides as before).  The layout amdnuwa_xattn2_bwd_ex writes its dS / Pm in
     * with flag bit 0.  amdnuwa_gemm_tn_chunked_a_supported() first; AMDNUWA_ERR_UNSUPPORTED otherwise. */
    int a_chunk32;
} amdnuwa_gemm_desc;

/* C[M,N] = alpha * A[M,K] . B[N,K]^T (+ bias).  K, lda, ldb multiples of 8. */
int amdnuwa_gemm_nt(const amdnuwa_gemm_desc* d, amdnuwa_stream stream);
/* 1 when the C2 (GEGLU) output of this product is produced inside the GEMM epilogue, 0 when the library will run GEMM + gate kernel */
int amdnuwa_gemm_nt_fused(const amdnuwa_gemm_desc* d);
int amdnuwa_gemm_nt_f16ops_supported(const amdnuwa_gemm_desc* d);
int amdnuwa_gemm_nt_f16x2_supported(const amdnuwa_gemm_desc* d);
/* 1 when amdnuwa_gemm_nt() honours d->c_lo_f16 for this product (it returns AMDNUWA_ERR_UNSUPPORTED otherwise) */
int amdnuwa_gemm_nt_f16_fused(const amdnuwa_gemm_desc* d);
/* out[r][c] = fp16(hi[r][c] + lo[r][c]) over an [R, C] view (row pitches ld_in / ld_out elements, C % 8 == 0) */
int amdnuwa_hilo_to_f16(const uint16_t* hi, const uint16_t* lo, int ld_in, uint16_t* out, int ld_out, long long R, int C, amdnuwa_stream stream);
/* C[M,N] (fp32) = beta*C + alpha * A[K,M]^T . B[K,N]   (reduction over the K token rows, split-K
 * through `workspace`, fixed summation order => deterministic).  lda, ldb multiples of 8; operand rows
 * must be readable up to the next multiple of 8 columns (padding content is irrelevant). */
/* 1 when amdnuwa_gemm_tn() takes this product with ab_f16 != 0 (fp16 operands: the four-wave kernel's shapes -- no token shift, token rows
 * a multiple of 64, both output dimensions >= 256); it returns AMDNUWA_ERR_UNSUPPORTED otherwise */
int amdnuwa_gemm_tn_f16_supported(const amdnuwa_gemm_desc* d);
/* 1 when amdnuwa_gemm_tn takes this product with A in planes of 32 columns (a_chunk32): batched, N <= 64, 128 < M <= 384 (the whole-M kernel) */
int amdnuwa_gemm_tn_chunked_a_supported(const amdnuwa_gemm_desc* d);
size_t amdnuwa_gemm_tn_workspace_bytes(const amdnuwa_gemm_desc* d);
int amdnuwa_gemm_tn(const amdnuwa_gemm_desc* d, void* workspace, size_t workspace_bytes, amdnuwa_stream stream);

/* ------------------------------------------------------------------------------------------
 * Row kernels: LayerNorm of SandwichNorm (np.py:112-128) fused with the residual add of
 * Transformer.forward (np.py:1175-1180); StableLayerNorm (np.py:88-95); GEGLU (np.py:255-258);
 * Embedding/AxialPositionalEmbedding/<bos> assemble (np.py:1659-1709, 1940-1944);
 * F.cross_entropy (np.py:1963).  fp32 statistics, bf16 hi[/lo] outputs where a GEMM consumes them.
 * ---------------------------------------------------------------------------------------- */
/* mode 0: out_hi[/lo] = LN(x)*w+b (bf16).  mode 1: out_f32 = resid + LN(x)*w+b.
 * stable != 0 (mode 0 only): x is first divided by its row amax (saved as 1/amax).
 * Input-type flags (fast bf16 mode keeps GEMM outputs in bf16): OR AMDNUWA_LN_X_BF16 into `mode` (ln_fwd) or into
 * `stable` (ln_bwd) when x points at bf16 values; OR AMDNUWA_LN_DY_BF16 into ln_bwd's `stable` when dy does. */
/* Token shift folded into the pre-norm's store / the pre-norm backward's reads (shift_ntok = rows per sample, 0 = none):
 * shift_fmap > 0: ShiftVideoTokens (np.py:185-253) on a fmap x fmap grid, <bos> row untouched; shift_fmap == -1: ShiftAudioTokens
 * (np.py:157-183): the first half of the channels of row i comes from row i - 1 (zeros for row 0), every row takes part. */
#define AMDNUWA_LN_X_BF16 16
#define AMDNUWA_LN_DY_BF16 32
/* ln_fwd (mode 0) / ln_post_pre_fwd: the second 16-bit output (out_lo / h_lo) receives the FP16 rendering of the normalised row
 * instead of the bf16 residual -- the A operand of the fp16-operand GEMMs (amdnuwa_gemm_desc.ab_f16) */
#define AMDNUWA_LN_LO_F16 64
/* Round 5, fp16 everywhere a block needs ONE 16-bit copy.  AMDNUWA_LN_OUT_F16 -- in ln_fwd's `mode` / ln_post_pre_fwd's `flags`: out_hi /
 * h_hi itself receives the FP16 rendering (saturating; out_lo / h_lo must be NULL): the block's forward GEMM, its weight-gradient GEMM and
 * nothing else read it.  In ln_bwd_f16's `stable`: dx_hi receives fp16(S * dx) (dx_lo NULL).  AMDNUWA_LN_DY_F16 (ln_bwd_f16's `stable`):
 * dy points at fp16(S * value).  S travels as a DEVICE pointer scale2 = {S, 1 / S} (NULL = 1): a power of two chosen once per backward
 * pass by the host side from the residual-stream gradient, without a host synchronisation (nuwa_pytorch_amd/ops.py: _grad_scale). */
#define AMDNUWA_LN_OUT_F16 128
#define AMDNUWA_LN_DY_F16 256
/* ln_bwd_f16's `stable`: the fp32 dy is multiplied by the device scalar scale2[1] on the way in (the upstream gradient of the loss entering the
 * final norm's backward: no separate scaling pass over dy) */
#define AMDNUWA_LN_DY_SCALED 512
/* ln_fwd's `mode` (with bit 0, the post-norm form): out_f32 = resid - LN(x) instead of resid + LN(x): a reversible block's input recomputed from
 * its output (rev.py:77-106: x2 = y2 - g(y1), x1 = y1 - f(x2)) without negation passes over the stream */
#define AMDNUWA_LN_RESID_MINUS 1024
int amdnuwa_ln_fwd(const float* x, const float* resid, const float* w, const float* b, uint16_t* out_hi,
                   uint16_t* out_lo, float* out_f32, float* mean, float* rstd, float* inv_amax, long long R, int D,
                   int mode, int stable, float eps, int shift_ntok, int shift_fmap, amdnuwa_stream stream);
/* shift_ntok > 0 (mode 0): the bf16 output is written THROUGH the forward token shift of ShiftVideoTokens (np.py:210-253), i.e.
 * out = shift(LN(x)): rows are tokens of samples of shift_ntok rows (<bos> first), shift_fmap = tokens per grid row / column */
/* Post-norm + residual of one block fused with the pre-norm (+ token shift) of the NEXT block (Transformer.forward runs the
 * blocks back to back, np.py:1175-1180): out_f32 = resid + LN(y; w, b) with (mean, rstd) saved as mode 1 does, then
 * h = shift(LN(out_f32; next_w, next_b)) in bf16 hi[/lo] with (next_mean, next_rstd) saved as mode 0 does.
 * flags: AMDNUWA_LN_X_BF16 when y points at bf16 values. */
int amdnuwa_ln_post_pre_fwd(const float* y, const float* resid, const float* w, const float* b, float* out_f32,
                            float* mean, float* rstd, const float* next_w, const float* next_b, uint16_t* h_hi,
                            uint16_t* h_lo, float* next_mean, float* next_rstd, long long R, int D, int flags, float eps,
                            int shift_ntok, int shift_fmap, amdnuwa_stream stream);
/* Chained backward across a block boundary (the mirror of amdnuwa_ln_post_pre_fwd): the pre-norm backward of block k+1,
 *   dx = g + dLN(dh; x, mean, rstd, w)        (dh read through the inverse token shift when shift_ntok > 0; dw, db its weight grads)
 * and, on the same row while it is in registers, the post-norm backward of block k,
 *   dy_prev = dLN(dx; y_prev, mean_prev, rstd_prev, w_prev)   (bf16 hi[/lo]; dw_prev, db_prev; dsum_prev = column sums, optional).
 * inputs_bf16: 0 = dh and y_prev point at fp32 values, 1 = both at bf16 values (the all-bf16 mode), 2 = dh bf16 and y_prev fp32
 * (the 'bf16x3-fwd' mode: fp32 post-norm inputs from its 3-MFMA forward, bf16 dgrad outputs in its backward). */
size_t amdnuwa_ln_bwd_chain_workspace_bytes(long long R, int D);
int amdnuwa_ln_bwd_chain(const void* dh, const float* x, const float* mean, const float* rstd, const float* w, const float* g,
                         float* dx, float* dw, float* db, const void* y_prev, const float* mean_prev, const float* rstd_prev,
                         const float* w_prev, uint16_t* dy_prev_hi, uint16_t* dy_prev_lo, float* dw_prev, float* db_prev,
                         float* dsum_prev, long long R, int D, int shift_ntok, int shift_fmap, int inputs_bf16, void* workspace,
                         size_t workspace_bytes, amdnuwa_stream stream);
/* the same with fp16 gradients: inputs_bf16 = 3 -> dh points at fp16(S * value) (y_prev fp32); dy_prev_f16 != 0 -> dy_prev_hi receives
 * fp16(S * dy_prev), saturating (dy_prev_lo NULL).  inputs_bf16 0..2 / dy_prev_f16 = 0 / scale2 = NULL is amdnuwa_ln_bwd_chain. */
int amdnuwa_ln_bwd_chain_f16(const void* dh, const float* x, const float* mean, const float* rstd, const float* w, const float* g,
                             float* dx, float* dw, float* db, const void* y_prev, const float* mean_prev, const float* rstd_prev,
                             const float* w_prev, uint16_t* dy_prev_hi, uint16_t* dy_prev_lo, float* dw_prev, float* db_prev,
                             float* dsum_prev, long long R, int D, int shift_ntok, int shift_fmap, int inputs_bf16, int dy_prev_f16,
                             const float* scale2, void* workspace, size_t workspace_bytes, amdnuwa_stream stream);
size_t amdnuwa_ln_bwd_workspace_bytes(long long R, int D);
/* dy fp32; shift_ntok > 0 reads dy through the inverse token shift.  Exactly one of dx_hi (bf16
 * hi[/lo] output) / dx_acc (fp32) is non-NULL; dx_acc = (dres ? dres : dx_acc) + dx.  dw, db, dsum
 * (= column sums of dx) may be NULL; accumulate != 0 adds into them. */
int amdnuwa_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* inv_amax,
                   const float* w, uint16_t* dx_hi, uint16_t* dx_lo, float* dx_acc, const float* dres, float* dw,
                   float* db, float* dsum, long long R, int D, int shift_ntok, int shift_fmap, int stable, int accumulate, void* workspace,
                   size_t workspace_bytes, amdnuwa_stream stream);
/* amdnuwa_ln_bwd with the fp16-gradient flags of `stable` (AMDNUWA_LN_DY_F16, AMDNUWA_LN_OUT_F16) and the device scale pair */
int amdnuwa_ln_bwd_f16(const float* dy, const float* x, const float* mean, const float* rstd, const float* inv_amax,
                       const float* w, uint16_t* dx_hi, uint16_t* dx_lo, float* dx_acc, const float* dres, float* dw,
                       float* db, float* dsum, long long R, int D, int shift_ntok, int shift_fmap, int stable, int accumulate,
                       const float* scale2, void* workspace, size_t workspace_bytes, amdnuwa_stream stream);
size_t amdnuwa_colsum_workspace_bytes(long long R, int D);
int amdnuwa_colsum(const float* x, float* out, long long R, int D, int accumulate, void* workspace,
                   size_t workspace_bytes, amdnuwa_stream stream);
/* u = [a | g], each FP columns wide: o = a * gelu_erf(g) */
int amdnuwa_geglu_fwd(const uint16_t* u_hi, const uint16_t* u_lo, uint16_t* o_hi, uint16_t* o_lo, long long R, int FP,
                      amdnuwa_stream stream);
/* the same on the interleaved-by-8 layout of u: columns [16 q, 16 q + 8) hold a_{8q..8q+7}, [16 q + 8, 16 q + 16) their gates
 * (FP % 8 == 0).  The FF1 weight rows are permuted accordingly, so a lane of the GEMM epilogue holds a value and its gate. */
int amdnuwa_geglu_il_fwd(const uint16_t* u_hi, const uint16_t* u_lo, uint16_t* o_hi, uint16_t* o_lo, long long R, int FP,
                         amdnuwa_stream stream);
int amdnuwa_geglu_il_bwd(const uint16_t* u_hi, const uint16_t* u_lo, const uint16_t* d_hi, const uint16_t* d_lo,
                         uint16_t* du_hi, uint16_t* du_lo, long long R, int FP, amdnuwa_stream stream);
int amdnuwa_geglu_bwd(const uint16_t* u_hi, const uint16_t* u_lo, const uint16_t* d_hi, const uint16_t* d_lo,
                      uint16_t* du_hi, uint16_t* du_lo, long long R, int FP, amdnuwa_stream stream);
/* dst[r][c<C] = bf16(src[r][c]), zero for C <= c < Cp */
int amdnuwa_cast_pad(const float* src, int ld_src, uint16_t* hi, uint16_t* lo, int ld_dst, long long R, int C, int Cp,
                     amdnuwa_stream stream);
/* dst[c][r] = bf16(src[r][c]) */
int amdnuwa_transpose_cast(const float* src, int ld_src, uint16_t* hi, uint16_t* lo, int ld_dst, int R, int C,
                           amdnuwa_stream stream);
int amdnuwa_embed_fwd(const long long* ids, const float* W, const float* ax1, const float* ax2, const float* ax3,
                      const float* bos, float* x, int B, int ntok, int D, int H, int Wd, float frac,
                      amdnuwa_stream stream);
size_t amdnuwa_embed_bwd_workspace_bytes(int ntok, int D);
/* sorted_ids / perm (both [B*(ntok-1)], optional): the token ids stably sorted and their original flat positions -> the
 * token-embedding gradient is summed in a fixed order (deterministic); NULL selects fp32 atomics */
int amdnuwa_embed_bwd(const long long* ids, const long long* sorted_ids, const long long* perm, const float* dx, float* dW,
                      float* dax1, float* dax2, float* dax3, float* dbos, int B, int ntok, int D, int F, int H, int Wd,
                      float frac, void* workspace, size_t workspace_bytes, amdnuwa_stream stream);
/* row_loss[r] = lse(logits[r]) - logits[r][t]; *loss = mean(row_loss);
 * dl (optional, bf16 hi[/lo], ld = ld_dl) = (softmax - onehot) * grad_scale */
int amdnuwa_ce_fwd(const float* logits, const long long* targets, float* row_loss, float* loss, uint16_t* dl_hi,
                   uint16_t* dl_lo, long long R, int C, int ld_dl, float grad_scale, amdnuwa_stream stream);
int amdnuwa_scale_by_device_scalar(float* x, size_t n, const float* scalar, amdnuwa_stream stream);
/* Fused to_logits + cross entropy (np.py:1958 `self.to_logits(...)` feeding np.py:1963 `F.cross_entropy`) WITHOUT the fp32 logits in
 * memory: logits = h[R,K] . w[C,K]^T (bf16 operands, fp32 accumulate) are produced twice inside the 256x256 MFMA ring -- pass 1 keeps
 * per-(row, 64-column block) (max, sum exp) and the target logit, pass 2 writes dlogits = (softmax - onehot) * grad_scale as bf16
 * [R, ld_dl] (skipped when dlogits is NULL).  row_loss[r] = lse - logit[target]; *loss = mean (fixed order).  A target outside
 * [0, C) makes its row loss (and the mean) NaN.  Needs C % 64 == 0, K % 32 == 0 (else AMDNUWA_ERR_UNSUPPORTED: use gemm_nt + ce_fwd). */
size_t amdnuwa_linear_ce_workspace_bytes(long long R, int C);
int amdnuwa_linear_ce(const uint16_t* h, int ldh, const uint16_t* w, int ldw, const long long* targets, long long R, int C, int K,
                      float grad_scale, float* row_loss, float* loss, uint16_t* dlogits, int ld_dl, void* workspace,
                      size_t workspace_bytes, amdnuwa_stream stream);
/* The same with both operands as bf16 hi + lo pairs: the to_logits of the 'bf16x3-fwd' mode, whose logits (and so the loss) must stay
 * inside the 1e-3 bound while its backward takes a plain bf16 dlogits.  Pass 1 (statistics, lse, loss) runs three MFMAs per product
 * on the hi + lo ring.  Pass 2 (dlogits): with h_f16 / w_f16 (fp16 renderings of the same operands, same leading dimensions; both or
 * neither) ONE fp16 MFMA per product against the exact lse of pass 1 -- the error of an fp16 product sits below the bf16 rounding of
 * dlogits itself; with both NULL the hi + lo ring again.  Same workspace, same outputs as amdnuwa_linear_ce. */
int amdnuwa_linear_ce_x3(const uint16_t* h_hi, const uint16_t* h_lo, const uint16_t* h_f16, int ldh, const uint16_t* w_hi,
                         const uint16_t* w_lo, const uint16_t* w_f16, int ldw, const long long* targets, long long R, int C, int K,
                         float grad_scale, float* row_loss, float* loss, uint16_t* dlogits, int ld_dl, void* workspace,
                         size_t workspace_bytes, amdnuwa_stream stream);

/* ------------------------------------------------------------------------------------------
 * Sparse3DNA core (np.py:488-608, incl. the unfoldNd gather np.py:526-534): causal (or symmetric) 3-D nearby
 * attention with <bos> key/value, fp32 softmax and talking heads.  q/k/v/o are token-row major
 * [B*ntok, ld] with head h in columns [h*dim_head, (h+1)*dim_head); q is UNSCALED.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int B, ntok;            /* ntok = 1 (<bos>) + number of video tokens present (<= F*H*W) */
    int F, H, W;            /* token grid (max_frames, fmap, fmap) */
    int kf, kh, kw;         /* kernel size */
    int df, dh, dw;         /* dilation */
    int heads, dim_head;    /* heads <= 8; dim_head in {32, 64}; W*heads*4 <= 512 */
    float scale;            /* dim_head ** -0.5 */
    /* relative-position bias (Sparse3DNA(rel_pos_bias=True), np.py:512-516, 542): rel_bias[j][head] is added to the score of
     * key slot j (j = 0 is <bos>: pass 0 there), fp32 [kf*kh*kw + 1][heads] or NULL.  The backward writes its gradient
     * (column sums of ds over every query) to d_rel_bias when that is non-NULL. */
    const float* rel_bias;
    float* d_rel_bias;
    /* 0: causal window -- every tap at or before the query on each axis (Sparse3DNA(causal=True), np.py:427);
     * 1: symmetric 'same' window, tap t of an axis at offset (t - (k-1)/2) * dilation (causal=False, np.py:429: the sketch
     *    encoder of NUWASketch).  Row 0 stays the <bos> key/value; taps that leave the grid are masked, taps inside the grid but
     *    past the end of the sequence read the reference's zero padding (score 0, value 0). */
    int noncausal;
} amdnuwa_s3_geom;

/* 1 when the window kernels (amdnuwa_sparse3dna_*, amdnuwa_cross2dna_*) take this geometry in the given operand form
 * (lo_operands != 0: bf16 hi + lo pairs): head size 32 / 64, <= 8 heads, W * heads * 4 <= 512, and the window's LDS tables
 * (they grow with J = kf*kh*kw + 1 key slots: Sparse3DNA / SparseCross2DNA kernel sizes, reference nuwa_pytorch.py:382-394,
 * 761-790) within the CU's 160 KiB for the forward AND both backward kernels.  The entry points themselves return
 * AMDNUWA_ERR_UNSUPPORTED for a geometry this query rejects. */
int amdnuwa_s3_supported(const amdnuwa_s3_geom* g, int lo_operands);
int amdnuwa_sparse3dna_fwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* k, const uint16_t* v,
                           const uint16_t* q_lo, const uint16_t* k_lo, const uint16_t* v_lo, int ld,
                           const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, amdnuwa_stream stream);
/* the forward core on SINGLE fp16 MFMAs (the MFMA band kernel with fp16 operands): q / k / v hold fp16 values (the second copy
 * amdnuwa_gemm_nt writes with c_lo_f16), o leaves as a bf16 hi + lo pair (o_lo may be NULL).  Geometry of the band kernels only
 * (causal window, 16-wide grid, 8 heads x 64, kw <= 3): amdnuwa_s3_f16_supported() says whether it applies.  The forward
 * Sparse3DNA core (reference nuwa_pytorch.py:488-608) of the 'bf16x3-fwd' mode. */
int amdnuwa_s3_f16_supported(const amdnuwa_s3_geom* g);
/* o_lo_f16 != 0: o_lo receives the FP16 rendering of the output instead of the bf16 residual -- the A operand of the two-MFMA to_out
 * product (amdnuwa_gemm_desc.ab_f16 with Blo); o stays the bf16 copy the backward reads. */
int amdnuwa_sparse3dna_fwd_f16(const amdnuwa_s3_geom* g, const uint16_t* q_f16, const uint16_t* k_f16, const uint16_t* v_f16, int ld,
                               const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, amdnuwa_stream stream);
size_t amdnuwa_sparse3dna_bwd_workspace_bytes(const amdnuwa_s3_geom* g);
int amdnuwa_sparse3dna_bwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* k, const uint16_t* v,
                           const uint16_t* q_lo, const uint16_t* k_lo, const uint16_t* v_lo, int ld,
                           const float* w_th, const uint16_t* dO, const uint16_t* dO_lo, int lddo, uint16_t* dq,
                           uint16_t* dk, uint16_t* dv, uint16_t* dq_lo, uint16_t* dk_lo, uint16_t* dv_lo, int ldd,
                           float* dw_th, int accumulate, void* workspace, size_t workspace_bytes,
                           amdnuwa_stream stream);
/* ABI 19, the fp16-gradient form of the backward (reference nuwa_pytorch.py:488-608 differentiated; block class 's' of the 'bf16x3-fwd' mode):
 * q / k / v are the fp16 arrays the fp16 forward read -- the block then keeps ONE 16-bit copy of them --, dO_f16 = fp16(S dO), and dq / dk / dv
 * leave as fp16(S gradient), saturating and counted by amdnuwa_f16_sat_count; dw_th leaves as fp32 WITHOUT the factor.  scale2 = device
 * pointer to {S, 1 / S} (S a power of two, see amdnuwa_gemm_desc.alpha_dev).  Same workspace as amdnuwa_sparse3dna_bwd.  Band-kernel
 * geometry without a relative-position bias only: amdnuwa_sparse3dna_bwd_f16_supported(), AMDNUWA_ERR_UNSUPPORTED otherwise.
 * amdnuwa_sparse3dna_fwd_f16 accepts o == NULL together with o_lo_f16 (the fp16 copy is then the only output). */
int amdnuwa_sparse3dna_bwd_f16_supported(const amdnuwa_s3_geom* g);
int amdnuwa_sparse3dna_bwd_f16(const amdnuwa_s3_geom* g, const uint16_t* q_f16, const uint16_t* k_f16, const uint16_t* v_f16, int ld,
                               const float* w_th, const uint16_t* dO_f16, int lddo, uint16_t* dq_f16, uint16_t* dk_f16, uint16_t* dv_f16,
                               int ldd, float* dw_th, int accumulate, const float* scale2, void* workspace, size_t workspace_bytes,
                               amdnuwa_stream stream);

/* SparseCross2DNA (np.py:761-901): NUWASketch's decoder cross-attention.  Queries = the video rows q [B*ntok, ldq] (row 0 of every
 * sample is <bos>); keys / values = the sketch context k, v [B*ctx_rows, ldkv], ctx_rows = g->kf * H * W (g->kf = sketch frames).
 * Query (f, y, x) attends to the learned null key / value (key slot 0; null_k / null_v: bf16 [heads*dim_head]) and, in EVERY context
 * frame a, to the g->kh x g->kw neighbourhood of (y, x) with dilation (g->dh, g->dw) and 'same' zero padding -- slot
 * 1 + (a*kh + b)*kw + c; key_mask [B][ctx_rows] (1 = visible, NULL = all) masks slots; fp32 softmax; talking heads w_th [heads][heads].
 * g->df, g->noncausal and g->rel_bias are ignored (rel_bias must be NULL).  Row 0 of every sample (the <bos> query attends to ALL
 * context tokens, without talking heads: np.py:813-830) is NOT read or written here: the caller owns o / dq rows b*ntok.
 * Backward: dq rows 1.., dk / dv [B*ctx_rows, lddkv], d_null_k / d_null_v fp32 [heads*dim_head], dw_th fp32 [heads*heads]. */
int amdnuwa_cross2dna_fwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq, int ctx_rows,
                          const uint16_t* k, const uint16_t* v, const uint16_t* k_lo, const uint16_t* v_lo, int ldkv,
                          const uint16_t* null_k, const uint16_t* null_k_lo, const uint16_t* null_v, const uint16_t* null_v_lo,
                          const uint8_t* key_mask, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, amdnuwa_stream stream);
size_t amdnuwa_cross2dna_bwd_workspace_bytes(const amdnuwa_s3_geom* g);
int amdnuwa_cross2dna_bwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq, int ctx_rows,
                          const uint16_t* k, const uint16_t* v, const uint16_t* k_lo, const uint16_t* v_lo, int ldkv,
                          const uint16_t* null_k, const uint16_t* null_k_lo, const uint16_t* null_v, const uint16_t* null_v_lo,
                          const uint8_t* key_mask, const float* w_th, const uint16_t* dO, const uint16_t* dO_lo, int lddo,
                          uint16_t* dq, uint16_t* dq_lo, int lddq, uint16_t* dk, uint16_t* dv, uint16_t* dk_lo, uint16_t* dv_lo,
                          int lddkv, float* d_null_k, float* d_null_v, float* dw_th, void* workspace, size_t workspace_bytes,
                          amdnuwa_stream stream);

/* ------------------------------------------------------------------------------------------
 * Incremental decoding for NUWA.generate (np.py:1841-1915).  The reference recomputes the whole prefix for every sampled
 * token; every decoder stage is causal, so row `pos` only needs cached rows < pos.  `pos` (row index inside each sample,
 * 0 = <bos>) is read from DEVICE memory so one captured HIP graph serves every token.
 * ---------------------------------------------------------------------------------------- */
/* ShiftVideoTokens (np.py:210-253) for the single new row: h [B, D] (bf16 hi[/lo]) is stored as row pos of
 * cache [B, cache_rows, D] and out [B, D] = shift(h)[pos] (first channel quarter from row pos - fmap, second from pos - 1,
 * zero at the frame border; <bos> unchanged). */
int amdnuwa_decode_shift(const uint16_t* h_hi, const uint16_t* h_lo, uint16_t* cache_hi, uint16_t* cache_lo,
                         uint16_t* out_hi, uint16_t* out_lo, const int* pos, int B, int cache_rows, int D, int fmap,
                         amdnuwa_stream stream);
/* The norms around a block for the one new row per sample, in one launch (SandwichNorm, np.py:112-128, + the residual add of
 * Transformer.forward, np.py:1175-1180):  x_new = resid + LN(y; w, b)  [skipped when resid == NULL: x_new = y, fp32];
 * h = LN(x_new; next_w, next_b)  [skipped when next_w == NULL];  with cache_hi != NULL h also becomes row pos of
 * cache [B, cache_rows, D] and out = shift(h)[pos] as amdnuwa_decode_shift does, else out = h.  y: fp32 or (y_is_bf16) bf16.
 * fmap = -1 selects ShiftAudioTokens (np.py:157-183): the first half of the channels comes from row pos - 1 (<bos> included). */
int amdnuwa_decode_ln(const void* y, int y_is_bf16, const float* resid, const float* w, const float* b, const float* next_w,
                      const float* next_b, float* x_new, uint16_t* cache_hi, uint16_t* cache_lo, uint16_t* out_hi,
                      uint16_t* out_lo, const int* pos, int B, int cache_rows, int D, int fmap, float eps,
                      amdnuwa_stream stream);
/* Sparse3DNA core (np.py:488-608) for the single new query: qkv [B, 3*inner] (q | k | v of row pos, q unscaled); its k | v
 * join kv_cache [B, cache_rows, 2*inner]; o [B, inner] = attention over <bos> + the causal taps read from the cache
 * (g->rel_bias as in amdnuwa_sparse3dna_fwd; g->ntok is ignored). */
int amdnuwa_s3_decode(const amdnuwa_s3_geom* g, const uint16_t* qkv, const uint16_t* qkv_lo, uint16_t* kv_cache,
                      uint16_t* kv_cache_lo, int cache_rows, const int* pos, const float* w_th, uint16_t* o,
                      uint16_t* o_lo, amdnuwa_stream stream);

/* ------------------------------------------------------------------------------------------
 * Text cross-attention core (Attention.forward with context, np.py:339-378): learned null key/value
 * at key slot 0, context key mask, fp32 softmax, talking heads, attn @ v.  q/o are token-row major
 * [B*n, ld]; keys/values are packed per (sample, head) by amdnuwa_xattn_pack.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int B, n, T;            /* queries per sample, context length */
    int JP;                 /* amdnuwa_xattn_jp(T): T+1 rounded up to a multiple of 32 (<= 288) */
    int heads, dim_head;    /* heads <= 8, dim_head in {32, 64} */
    float scale;
} amdnuwa_xattn_geom;

typedef struct {            /* all [B][heads][JP][dim_head] (Kp, Vp) or [B][heads][dim_head][JP] (Kt, Vt), bf16 */
    uint16_t *Kp, *Kp_lo, *Kt, *Kt_lo, *Vp, *Vp_lo, *Vt, *Vt_lo;
    uint8_t* valid;         /* [B][JP] */
} amdnuwa_xattn_kv;

int amdnuwa_xattn_jp(int T);
/* kv: [B*T, ldkv] bf16, keys in columns [0, inner), values in [inner, 2*inner) (= to_kv(context)); ldkv % 8 == 0 and 16-byte
 * aligned bases (the kernels move 16-byte pieces), AMDNUWA_ERR_ARG otherwise -- the same holds for amdnuwa_xattn_unpack */
int amdnuwa_xattn_pack(const amdnuwa_xattn_geom* g, const uint16_t* kv, const uint16_t* kv_lo, int ldkv,
                       const float* null_k, const float* null_v, const uint8_t* context_mask,
                       const amdnuwa_xattn_kv* packed, amdnuwa_stream stream);
/* the same with kv_f16 = the fp16 rendering of to_kv(context) in place of the bf16 residuals: the *_lo images of `packed` become fp16
 * images (what amdnuwa_xattn2_fwd_f16 reads); the hi images stay the bf16 ones the backward kernels read */
int amdnuwa_xattn_pack_f16(const amdnuwa_xattn_geom* g, const uint16_t* kv, const uint16_t* kv_f16, int ldkv,
                           const float* null_k, const float* null_v, const uint8_t* context_mask,
                           const amdnuwa_xattn_kv* packed, amdnuwa_stream stream);
/* P / Pm (optional): softmax probabilities before / after talking heads, [B][heads][n][JP] bf16 hi[/lo],
 * saved for the backward */
int amdnuwa_xattn_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq,
                      const amdnuwa_xattn_kv* packed, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo,
                      uint16_t* P, uint16_t* P_lo, uint16_t* Pm, uint16_t* Pm_lo, amdnuwa_stream stream);
/* the same, plus (optional) the softmax statistics [B][heads][n][2] fp32 = (row max of the scaled, masked scores in the log2
 * domain, 1 / row sum of exp): what amdnuwa_xattn2_bwd recomputes the probabilities from.  This is how the 'bf16x3-fwd' mode
 * pairs the 3-MFMA forward (Attention.forward, reference nuwa_pytorch.py:339-378) with the recomputing bf16 backward: P / Pm may
 * then be NULL and nothing of size n x JP is saved */
int amdnuwa_xattn_fwd_stats(const amdnuwa_xattn_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq,
                            const amdnuwa_xattn_kv* packed, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo,
                            uint16_t* P, uint16_t* P_lo, uint16_t* Pm, uint16_t* Pm_lo, float* stats, amdnuwa_stream stream);
size_t amdnuwa_xattn_bwd_workspace_bytes(const amdnuwa_xattn_geom* g);
/* query side of the backward: dq and ds = dL/dsim [B][heads][n][JP]; dK/dV follow as batched
 * amdnuwa_gemm_tn over ds / Pm, then amdnuwa_xattn_unpack */
int amdnuwa_xattn_bwd(const amdnuwa_xattn_geom* g, const uint16_t* dO, const uint16_t* dO_lo, int lddo,
                      const amdnuwa_xattn_kv* packed, const float* w_th, const uint16_t* P, const uint16_t* P_lo,
                      uint16_t* dS, uint16_t* dS_lo, uint16_t* dq, uint16_t* dq_lo, int lddq, float* dw_th,
                      int accumulate, void* workspace, size_t workspace_bytes, amdnuwa_stream stream);
/* accumulate: bit 0 = add into dnull_k / dnull_v instead of overwriting them; bit 1 = the rows of dKp / dVp follow the chunk-permuted
 * key order in which amdnuwa_xattn2_bwd writes its dS / Pm columns (key 32 c + kk at row 32 c + 8 ((kk & 15) >> 2) + 4 (kk >> 4) + (kk & 3):
 * the order a lane of the kernel holds its 8 slots, one 16-byte store each); amdnuwa_xattn_bwd and amdnuwa_xattn2_bwd_rc use the plain order;
 * bit 2 (ABI 18) = key order of amdnuwa_xattn6_bwd: context key t at position t, the null key at position T (else: null key 0, context key t at t + 1) */
int amdnuwa_xattn_unpack(const amdnuwa_xattn_geom* g, const float* dKp, const float* dVp, uint16_t* dkv,
                         uint16_t* dkv_lo, int ldkv, float* dnull_k, float* dnull_v, int accumulate,
                         amdnuwa_stream stream);

/* ---- cross-attention core, second design (fast bf16 mode only: heads == 8, dim_head == 64, T + 1 <= 288) --------------
 * One wave = 16 queries x all heads, head mix in registers, K/V chunks shared through a direct-to-LDS ring; the forward saves
 * only the softmax statistics stats[B][heads][n][2] = (row max, 1 / row sum) and the backward recomputes the probabilities.
 * _bwd writes dS and Pm ([B][heads][n][JP] bf16, as amdnuwa_xattn_bwd / _fwd do) for the batched dK / dV GEMMs, dq, and the
 * per-workgroup talking-heads partials part_th (reduce with amdnuwa_colsum or the Python side). */
int amdnuwa_xattn2_supported(const amdnuwa_xattn_geom* g);
int amdnuwa_xattn2_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const amdnuwa_xattn_kv* packed,
                       const float* w_th, uint16_t* o, int ldo, float* stats, amdnuwa_stream stream);
/* the same core on SINGLE fp16 MFMAs: q_f16 [B*n, ldq] and the *_lo images of `packed` (written by amdnuwa_xattn_pack_f16) hold fp16
 * values; o leaves as a bf16 hi + lo pair (o_lo may be NULL).  The forward cross-attention core of the 'bf16x3-fwd' mode: with
 * the hi + lo projection GEMMs around it the full-depth logits stay within 1e-3 of the fp32 reference (tools/error_budget.py) */
/* o_lo_f16: as for amdnuwa_sparse3dna_fwd_f16 */
int amdnuwa_xattn2_fwd_f16(const amdnuwa_xattn_geom* g, const uint16_t* q_f16, int ldq, const amdnuwa_xattn_kv* packed,
                           const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, float* stats, amdnuwa_stream stream);
size_t amdnuwa_xattn2_bwd_workspace_bytes(const amdnuwa_xattn_geom* g);
/* dS / Pm [B][heads][n][JP]: the keys of every 32-key chunk in the order the kernel's lanes hold them (position 8 g + e of a chunk = key
 * (e < 4 ? 4 g + e : 16 + 4 g + e - 4)): batched amdnuwa_gemm_tn over them yields dKp / dVp rows in the same order, which
 * amdnuwa_xattn_unpack undoes when its flag bit 1 is set */
int amdnuwa_xattn2_bwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                       const amdnuwa_xattn_kv* packed, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm,
                       uint16_t* dq, int lddq, float* part_th, size_t part_bytes, amdnuwa_stream stream);
/* ABI 17: the same with flags.  Bit 0: dS / Pm are written CHUNK-MAJOR, [B][heads][JP / 32][n][32] (same chunk-permuted key order inside a
 * chunk): a store instruction of the kernel then covers 1 KiB of contiguous memory (16 queries x 64 B) instead of sixteen 64-byte pieces at
 * the 2 JP-byte row pitch (xattn3_bwd 2035 -> 1878 us at b = 128).  Read them with amdnuwa_gemm_tn and a_chunk32.  AMDNUWA_ERR_UNSUPPORTED when
 * the shape is not supported. */
int amdnuwa_xattn2_bwd_ex(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                          const amdnuwa_xattn_kv* packed, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm,
                          uint16_t* dq, int lddq, float* part_th, size_t part_bytes, int flags, amdnuwa_stream stream);
/* The recomputing backward without the dS / Pm arrays (n % 32 == 0): the query side as above (dq, part_th) leaves nb / delta per
 * (head, query) in `nbd` (>= amdnuwa_xattn2_bwd_rc_stats_bytes), and a key-side kernel -- one workgroup per 32 keys of a sample,
 * the queries streamed through LDS -- recomputes ds and P' from them and accumulates dKp / dVp ([B][heads][JP][dim_head] fp32,
 * dKp already scaled: the inputs of amdnuwa_xattn_unpack).  Same rounding points as _bwd + the two batched amdnuwa_gemm_tn;
 * 3 GB less written and read back per layer call at b = 128. */
int amdnuwa_xattn2_bwd_rc_supported(const amdnuwa_xattn_geom* g);
size_t amdnuwa_xattn2_bwd_rc_stats_bytes(const amdnuwa_xattn_geom* g);
int amdnuwa_xattn2_bwd_rc(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                          const amdnuwa_xattn_kv* packed, const float* w_th, const float* stats, uint16_t* dq, int lddq,
                          float* part_th, size_t part_bytes, float* nbd, size_t nbd_bytes, float* dKp, float* dVp,
                          amdnuwa_stream stream);
/* ---- cross-attention core, third design ("xattn6", ABI 18; reference nuwa_pytorch.py:339-378): heads == 8, dim_head == 64, ANY context
 * length T >= 1 (g->JP is ignored).  Keys / values travel as images in the kernels' LDS order, written once per layer call by
 * amdnuwa_xattn6_pack from the 16-bit to_kv(context) rows: K6 / V6 [B][nch][heads][32][64] 16-bit with nch = amdnuwa_xattn6_nch(T) =
 * 2 * ceil(T / 64) chunks of 32 keys (K [key][d]; V transposed, key slots in the lanes' order; bank swizzles baked in),
 * vbits [B][nch]: bit j of word c = context key 32 c + j exists and passes context_mask.  The learned null key / value (np.py:343-347) are
 * NOT image rows: the kernels take null_k / null_v [heads][dim_head] fp32 and treat the null key as a rank-one term.
 * f16 != 0: q16 / kv16 and the images are fp16 and every MFMA is the fp16 one ('bf16x3-fwd'); else bf16.
 * stats [B][heads][n][2] as amdnuwa_xattn2_fwd writes them (any reference maximum m in the log2 domain, 1 / sum of exp2(s - m) incl. the
 * null key): what amdnuwa_xattn2_bwd* recompute the probabilities from.  o / o_lo / o_lo_f16 as amdnuwa_xattn2_fwd_f16. */
typedef struct { uint16_t *K6, *V6; uint32_t* vbits; } amdnuwa_xattn6_kv;
int amdnuwa_xattn6_supported(const amdnuwa_xattn_geom* g);
int amdnuwa_xattn6_nch(int T);
size_t amdnuwa_xattn6_image_bytes(const amdnuwa_xattn_geom* g);      /* bytes of K6 (and of V6); vbits: B * nch words */
int amdnuwa_xattn6_pack(const amdnuwa_xattn_geom* g, const uint16_t* kv16, int ldkv, const uint8_t* context_mask, int f16,
                        const amdnuwa_xattn6_kv* out, amdnuwa_stream stream);
int amdnuwa_xattn6_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q16, int ldq, const amdnuwa_xattn6_kv* kv, const float* null_k,
                       const float* null_v, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, float* stats,
                       int f16, amdnuwa_stream stream);
/* The query side of the recomputing backward on the same recipe (bf16; replaces amdnuwa_xattn2_bwd_ex with flag bit 0 where g->JP / 32 <= 64):
 * amdnuwa_xattn6_pack_bwd writes K6 / V6 [B][JP / 32][heads][32][64] bf16 ([key][d] tiles in LDS order; positions 0..T-1 = the context keys,
 * position T = the null key: the key order of its dS / Pm, which amdnuwa_xattn_unpack reads with flag bit 2 set) and vbits [B][JP / 32]; amdnuwa_xattn6_bwd takes them, q / dO [B*n, ld] bf16
 * and the forward's statistics and writes dq, dS / Pm CHUNK-MAJOR ([B][heads][JP / 32][n][32], chunk-permuted slots, lane groups without a
 * key unwritten -- exactly amdnuwa_xattn2_bwd_ex's flag-bit-0 layout) and the talking-heads partials part_th (>= _workspace_bytes). */
size_t amdnuwa_xattn6_bwd_image_bytes(const amdnuwa_xattn_geom* g);
int amdnuwa_xattn6_pack_bwd(const amdnuwa_xattn_geom* g, const uint16_t* kv, int ldkv, const float* null_k, const float* null_v,
                            const uint8_t* context_mask, const amdnuwa_xattn6_kv* out, amdnuwa_stream stream);
size_t amdnuwa_xattn6_bwd_workspace_bytes(const amdnuwa_xattn_geom* g);
int amdnuwa_xattn6_bwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo, const amdnuwa_xattn6_kv* kv,
                       const float* null_k, const float* null_v, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm, uint16_t* dq, int lddq, float* part_th,
                       size_t part_bytes, amdnuwa_stream stream);
/* ABI 19, the fp16-gradient form (block class 'x' of the 'bf16x3-fwd' mode): images from the FP16 copy of to_kv(context)
 * (amdnuwa_xattn6_pack_bwd_f16; the null key / value rounded to fp16), q = the fp16 copy the forward read, dO = fp16(S dO); dq leaves as
 * fp16(S dq), saturating and counted by amdnuwa_f16_sat_count, dS as fp16(S dS) and Pm as fp16 through the plain converter; the V image carries the
 * factor 2^-6 (the kernel's dP' = dO . V then stays inside fp16 where dO does) and the part_th partials the factor S / 64.  Every MFMA is
 * the fp16 one.  The dK / dV products over dS / Pm then run amdnuwa_gemm_tn with ab_f16 + a_chunk32 and alpha_dev = 1 / S.
 * amdnuwa_xattn6_fwd accepts o == NULL together with o_lo_f16 (the fp16 copy is then the only output). */
int amdnuwa_xattn6_pack_bwd_f16(const amdnuwa_xattn_geom* g, const uint16_t* kv_f16, int ldkv, const float* null_k, const float* null_v,
                                const uint8_t* context_mask, const amdnuwa_xattn6_kv* out, amdnuwa_stream stream);
int amdnuwa_xattn6_bwd_f16(const amdnuwa_xattn_geom* g, const uint16_t* q_f16, int ldq, const uint16_t* dO_f16, int lddo, const amdnuwa_xattn6_kv* kv,
                           const float* null_k, const float* null_v, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm, uint16_t* dq,
                           int lddq, float* part_th, size_t part_bytes, amdnuwa_stream stream);
/* Text cross-attention (Attention.forward with context, np.py:339-378) for ONE query row per sample (g->n must be 1):
 * q [B, ldq] unscaled, keys / values as packed by amdnuwa_xattn_pack (Kp / Vp images and the valid map), o [B, ldo]. */
int amdnuwa_xattn_decode(const amdnuwa_xattn_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq,
                         const amdnuwa_xattn_kv* packed, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo,
                         amdnuwa_stream stream);

/* ---- frozen VQGanVAE tokenizer (VQGanVAE.get_video_indices -> encode, reference vqgan_vae.py:431-435, 452-458), exact fp32 ---- */
typedef struct {
    int N, Cin, H, W, Cout, KH, KW, stride, pad;
    int Ho, Wo;             /* must equal (H + 2*pad - KH) / stride + 1 etc. */
    int leaky;              /* fuse LeakyReLU(0.1) (vqgan_vae.py:94-95) */
} amdnuwa_conv_desc;
/* nn.Conv2d forward, NCHW fp32, square stride/padding (replaces the encoders' convs, vqgan_vae.py:352-365, 228-242) */
int amdnuwa_conv2d_fwd(const amdnuwa_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                       amdnuwa_stream stream);
/* nn.GroupNorm(groups, C) [+ LeakyReLU(0.1)] on [N][C][HW] fp32 (ResBlock, vqgan_vae.py:233-237) */
int amdnuwa_groupnorm_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int HW, int groups,
                          float eps, int leaky, amdnuwa_stream stream);
/* cosine-similarity code lookup: indices[r] = argmax_c <l2norm(x[r]), l2norm(codebook[c])>, lowest index on exact ties
 * (eval path of vector_quantize_pytorch.VectorQuantize as called at vqgan_vae.py:368-378, 433); best_sim optional */
int amdnuwa_vq_argmax(const float* x, const float* codebook, long long* indices, float* best_sim, long long R,
                      int n_codes, int code_dim, amdnuwa_stream stream);
/* the same with a workspace (inverse code norms + per-slice partial arg-max): code_dim 256 runs a kernel that keeps the rows in
 * registers as MFMA operands and cuts the code axis into slices; other widths fall through to amdnuwa_vq_argmax */
size_t amdnuwa_vq_argmax_workspace_bytes(long long R, int n_codes);
int amdnuwa_vq_argmax_ws(const float* x, const float* codebook, long long* indices, float* best_sim, long long R, int n_codes,
                         int code_dim, void* workspace, size_t workspace_bytes, amdnuwa_stream stream);

/* VQGanAttention block of the encoder (vqgan_vae.py:244-286), exact fp32: in-place l2 normalisation of rows (q and k over the
 * spatial axis), the per-(image, head) attention core with the continuous-position bias [heads][P][P] precomputed from the
 * module's parameters and the learned log-scale, and LayerNormChan (+ residual) over the channel axis of an NCHW tensor. */
/* rows: `groups` groups of rows_per_group consecutive rows of length len, group g starting at row g * group_stride_rows */
int amdnuwa_rows_l2norm(float* x, int groups, int rows_per_group, int group_stride_rows, int len, amdnuwa_stream stream);
int amdnuwa_vqattn_core(const float* qkv, const float* bias, const float* scale, float* out, int N, int heads, int dim_head,
                        int P, amdnuwa_stream stream);
int amdnuwa_chan_layernorm(const float* x, const float* g, const float* b, const float* resid, float* y, int N, int C, int HW,
                           float eps, amdnuwa_stream stream);

/* VQGanVAE.decode pieces (vqgan_vae.py:437-441, GLUResBlock 212-226): nn.GLU over the channel axis of [N][2C][HW] and the
 * x2 bilinear nn.Upsample (align_corners=False) of an NCHW tensor; the rest of the decoder reuses conv2d / groupnorm / vqattn */
int amdnuwa_glu_chan(const float* x, float* y, int N, int C, int HW, amdnuwa_stream stream);
int amdnuwa_upsample_bilinear2x(const float* x, float* y, int N, int C, int H, int W, amdnuwa_stream stream);

/* ---- optimiser step of the trainer (row f2; reference train_nuwa.py:253-255 + optimizer.py:6-31) -------------------------------
 * "multi-tensor apply": a DEVICE table of chunks (<= 65536 elements each is a good size), every chunk pointing into one fp32
 * parameter / gradient / first- / second-moment tensor.  g == NULL marks a parameter without gradient this step (skipped). */
typedef struct {
    float* p; const float* g; float* m; float* v;
    long long n;
    float weight_decay;     /* 0 for the ndim < 2 parameters (optimizer.py:6-9) */
    float bias_correction1, bias_correction2;   /* 1 - beta^t with t = number of updates THIS parameter has received (torch keeps
                                                    the step count per parameter: one skipped for lack of a gradient lags behind) */
} amdnuwa_adamw_chunk;
/* out2[0] = global L2 norm of all gradients, out2[1] = min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0); fixed-order
 * reduction; partials: nchunks floats of scratch.  Everything stays on the device. */
int amdnuwa_grad_norm(const amdnuwa_adamw_chunk* chunks, int nchunks, float max_norm, float* partials, float* out2,
                      amdnuwa_stream stream);
int amdnuwa_scale_grads(const amdnuwa_adamw_chunk* chunks, int nchunks, const float* coef, amdnuwa_stream stream);
/* torch.optim.AdamW semantics (decoupled decay); gradients are multiplied by *clip_coef (device scalar, may be NULL) on the fly */
int amdnuwa_adamw_step(const amdnuwa_adamw_chunk* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                       const float* clip_coef, amdnuwa_stream stream);

/* ---- gradient exchange of the data-parallel step on RCCL (one process per GPU; SURVEY.md section 8 row e) --------------------
 * The reference has no multi-GPU code of its own (train_nuwa.py runs one device); this is what nuwa_pytorch_amd/distributed.py's
 * GradReducer(collective='native') puts under its flat fp32 buckets instead of torch.distributed.  librccl is dlopen()ed at the
 * first call (amdnuwa_comm_available() == 0 and AMDNUWA_ERR_UNSUPPORTED without it).  Rank 0 draws the 128-byte id, the host side
 * hands it to every rank (any channel: a torch.distributed object broadcast, a file, MPI), every rank calls _init on ITS device.
 * All collectives are in place, enqueued on `stream`, and return before the data moved; AMDNUWA_ERR_COMM = an RCCL error
 * (amdnuwa_comm_last_error() has its text, per thread). */
#define AMDNUWA_COMM_ID_BYTES 128
#define AMDNUWA_ERR_COMM -4
typedef struct amdnuwa_comm amdnuwa_comm;
int amdnuwa_comm_available(void);
const char* amdnuwa_comm_last_error(void);
int amdnuwa_comm_unique_id(void* id_out, size_t bytes);
int amdnuwa_comm_init(amdnuwa_comm** out, const void* id, size_t id_bytes, int rank, int world, int device);
int amdnuwa_comm_rank(const amdnuwa_comm* c);
int amdnuwa_comm_world(const amdnuwa_comm* c);
/* buf[0..count) <- sum (average != 0: mean, ncclAvg -- no separate division pass) over the ranks */
int amdnuwa_comm_allreduce(amdnuwa_comm* c, float* buf, size_t count, int average, amdnuwa_stream stream);
/* the same result over a store of world * shard floats as reduce-scatter + all-gather (rank r reduces buf + r * shard) */
int amdnuwa_comm_reduce_scatter_allgather(amdnuwa_comm* c, float* buf, size_t shard, int average, amdnuwa_stream stream);
/* bytes of buf <- rank `root`'s (the initial parameter broadcast) */
int amdnuwa_comm_broadcast(amdnuwa_comm* c, void* buf, size_t bytes, int root, amdnuwa_stream stream);
int amdnuwa_comm_destroy(amdnuwa_comm* c);

#ifdef __cplusplus
}
#endif
#endif
