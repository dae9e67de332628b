// HBM-bound row kernels of the decoder path (gfx950).  One 64-lane wave per token row, 16-byte
// accesses, fp32 statistics.  Replaces: nn.LayerNorm inside SandwichNorm (np.py:112-128),
// StableLayerNorm (np.py:88-95), the residual adds of Transformer.forward (np.py:1175-1180),
// GEGLU (np.py:255-258), Embedding + AxialPositionalEmbedding + <bos> concat (np.py:1659-1709,
// 1940-1944), F.cross_entropy (np.py:1963), and their autograd backward.
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

constexpr int MAXV = 4;        // float4 slots per lane: supports D <= 1024
constexpr int ROWS_PER_BLOCK = 4;

template <int NV> struct RowValsT { float4 v[NV]; };
typedef RowValsT<MAXV> RowVals;

// 4 consecutive elements starting at element index i of an fp32 (BF = false) or bf16 (BF = true) array
template <bool BF>
__device__ __forceinline__ float4 ld4(const void* base, size_t i) {
    if (BF) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + i);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + i);
}
// the same for an fp16 array
__device__ __forceinline__ float4 ld4h(const void* base, size_t i) {
    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + i);
    return make_float4(f16lo_f(v.x), f16hi_f(v.x), f16lo_f(v.y), f16hi_f(v.y));
}
// FMT: 0 fp32, 1 bf16, 2 fp16
template <int FMT>
__device__ __forceinline__ float4 ld4f(const void* base, size_t i) {
    if constexpr (FMT == 2) return ld4h(base, i);
    else return ld4<FMT == 1>(base, i);
}
template <int FMT, int NV>
__device__ __forceinline__ void row_load_f(const void* base, size_t row_off, int D, int lane, RowValsT<NV>& r, float mul = 1.f) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        const bool in = e < D;                       // branch-free (see row_load_t)
        const float4 v = ld4f<FMT>(base, row_off + (in ? e : 0));
        r.v[it] = make_float4(in ? v.x * mul : 0.f, in ? v.y * mul : 0.f, in ? v.z * mul : 0.f, in ? v.w * mul : 0.f);
    }
}
template <bool BF, int NV>
__device__ __forceinline__ void row_load_t(const void* base, size_t row_off, int D, int lane, RowValsT<NV>& r, float fill = 0.f) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        // branch-free: a load under a branch sits in its own basic block, the compiler's wait-count pass then loses the in-order count
        // of the loads in flight and every later use waits with vmcnt(0) -- the row kernels' loads went out one at a time (seen in the
        // ISA of the chained backward: GL [vmcnt(0)] GL [vmcnt(0)] ...).  Lanes past D re-read element 0 and select the fill.
        const bool in = e < D;
        const float4 v = ld4<BF>(base, row_off + (in ? e : 0));
        r.v[it] = make_float4(in ? v.x : fill, in ? v.y : fill, in ? v.z : fill, in ? v.w : fill);
    }
}

// non-temporal (streaming) forms: the row kernels touch every byte once, so the lines need not displace reusable data in L2 / MALL
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
template <bool BF>
__device__ __forceinline__ float4 ld4_nt(const void* base, size_t i) {
    if (BF) {
        const u32x2v v = __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(reinterpret_cast<const bf16_t*>(base) + i));
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
    const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(reinterpret_cast<const float*>(base) + i));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
    const f32x4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4v*>(p));
}
__device__ __forceinline__ void st_bf16x4_nt(bf16_t* p, float a, float b, float c, float d) {
    const u32x2v t = {pack2_rne(a, b), pack2_rne(c, d)};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x2v*>(p));
}

template <int NV>
__device__ __forceinline__ void row_load(const float* __restrict__ p, int D, int lane, RowValsT<NV>& r, float fill = 0.f) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        const bool in = e < D;                       // branch-free (see row_load_t)
        const float4 v = *reinterpret_cast<const float4*>(p + (in ? e : 0));
        r.v[it] = make_float4(in ? v.x : fill, in ? v.y : fill, in ? v.z : fill, in ? v.w : fill);
    }
}
// float4 at p + e when e < D, zeros otherwise -- without a branch (parameter vectors: weights, biases)
__device__ __forceinline__ float4 ldp4(const float* __restrict__ p, int e, int D) {
    const bool in = e < D;
    const float4 v = *reinterpret_cast<const float4*>(p + (in ? e : 0));
    return make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
}
// lo_f16: the second copy is the fp16 rendering of the value (operand of the fp16 GEMMs of 'bf16x3-fwd'), not the bf16 residual
// lo_f16 == 2 (lo must be NULL): `hi` itself receives the fp16 rendering -- the ONE 16-bit copy of the value when the backward of the block
// runs on fp16 operands as well (round 5: no bf16 copy next to it).  sat: running max |value| of the saturating stores (f16_sat_commit).
__device__ __forceinline__ void store_bf16x4(bf16_t* hi, bf16_t* lo, int e, float a, float b, float c, float d, int lo_f16 = 0, float* sat = nullptr) {
    float dummy = 0.f;
    float& mx = sat ? *sat : dummy;
    if (lo_f16 == 2) { *reinterpret_cast<uint2*>(hi + e) = make_uint2(pack2_f16_sat_n(a, b, mx), pack2_f16_sat_n(c, d, mx)); return; }
    if (!lo) { *reinterpret_cast<uint2*>(hi + e) = make_uint2(pack2_rne(a, b), pack2_rne(c, d)); return; }
    if (lo_f16) {
        *reinterpret_cast<uint2*>(hi + e) = make_uint2(pack2_rne(a, b), pack2_rne(c, d));
        *reinterpret_cast<uint2*>(lo + e) = make_uint2(pack2_f16_sat_n(a, b, mx), pack2_f16_sat_n(c, d, mx));
        return;
    }
    bf16_t h[4], l[4];
    f2bf_hilo(a, h[0], l[0]); f2bf_hilo(b, h[1], l[1]); f2bf_hilo(c, h[2], l[2]); f2bf_hilo(d, h[3], l[3]);
    *reinterpret_cast<uint2*>(hi + e) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    if (lo) *reinterpret_cast<uint2*>(lo + e) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

// Pre-norm store with the optional FORWARD token shift folded in (ShiftVideoTokens, np.py:210-253): the first quarter of the
// channels of token (f, y, w) belongs to token (f, y+1, w), the second to (f, y, w+1); a row writes zeros into its own quarter
// when it has no source (y == 0 / w == 0).  The consumers (NT and TN GEMMs) then read a plain matrix.
__device__ __forceinline__ void store_ln_shifted(bf16_t* out_hi, bf16_t* out_lo, long long row, int e, int D, int shift_ntok,
                                                 int shift_fmap, float y0, float y1, float y2, float y3, int lo_f16 = 0, float* sat = nullptr) {
    long long drow = row;
    bool keep = true, zero_own = false;
    if (shift_ntok > 0 && shift_fmap < 0) {
        // ShiftAudioTokens (np.py:157-183): the first HALF of the channels of token i belongs to token i + 1; every row takes part
        // (<bos> included) and row 0 keeps zeros there
        const int i = (int)(row % shift_ntok);
        if (e < (D >> 1)) { keep = i + 1 < shift_ntok; drow = row + 1; zero_own = i == 0; }
    } else if (shift_ntok > 0) {
        const int i = (int)(row % shift_ntok);
        const int qd = e < (D >> 2) ? 0 : (e < (D >> 1) ? 1 : 2);            // (no integer division in the row kernels)
        if (i > 0 && qd < 2) {
            const int p = i - 1, wq = p % shift_fmap, yq = (p / shift_fmap) % shift_fmap;
            if (qd == 0) { keep = yq + 1 < shift_fmap && i + shift_fmap < shift_ntok; drow = row + shift_fmap; zero_own = yq == 0; }
            else         { keep = wq + 1 < shift_fmap && i + 1 < shift_ntok;          drow = row + 1;          zero_own = wq == 0; }
        }
    }
    if (keep) store_bf16x4(out_hi + drow * D, out_lo ? out_lo + drow * D : nullptr, e, y0, y1, y2, y3, lo_f16, sat);
    if (zero_own) store_bf16x4(out_hi + row * D, out_lo ? out_lo + row * D : nullptr, e, 0.f, 0.f, 0.f, 0.f, lo_f16, sat);
}

template <int NV>
__device__ __forceinline__ void row_mean_rstd(const RowValsT<NV>& xv, int D, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) s += (xv.v[it].x + xv.v[it].y) + (xv.v[it].z + xv.v[it].w);
    mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        if (e < D) {
            const float a = xv.v[it].x - mean, b_ = xv.v[it].y - mean, c = xv.v[it].z - mean, d = xv.v[it].w - mean;
            q += (a * a + b_ * b_) + (c * c + d * d);
        }
    }
    rstd = rsqrtf(wave_sum(q) / D + eps);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward.
//   MODE 0 (pre-norm)        : out_bf16[hi,lo] = LN(x) * w + b
//   MODE 1 (post-norm + res) : out_f32 = resid + LN(x) * w + b
//   STABLE                   : x <- x / amax(x) first (StableLayerNorm np.py:93-95), saves 1/amax
// saves mean / rstd per row for the backward.
// ---------------------------------------------------------------------------------------------
template <int MODE, bool STABLE, int NV, bool XBF>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                     float* __restrict__ out_f32, float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, float* __restrict__ inv_amax_o,
                                                     long long R, int D, float eps, int shift_ntok, int shift_fmap, int lo_f16) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= R) return;
    RowValsT<NV> xv;
    row_load_t<XBF>(x, (size_t)row * D, D, lane, xv, STABLE ? -3.0e38f : 0.f);
    float inv_amax = 1.f;
    if (STABLE) {
        float m = -3.0e38f;
#pragma unroll
        for (int it = 0; it < NV; ++it) m = fmaxf(m, fmaxf(fmaxf(xv.v[it].x, xv.v[it].y), fmaxf(xv.v[it].z, xv.v[it].w)));
        m = wave_max(m);
        inv_amax = 1.f / m;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            if (e < D) { xv.v[it].x = xv.v[it].x / m; xv.v[it].y = xv.v[it].y / m; xv.v[it].z = xv.v[it].z / m; xv.v[it].w = xv.v[it].w / m; }
            else xv.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float mean, rstd;
    row_mean_rstd<NV>(xv, D, lane, eps, mean, rstd);
    if (lane == 0) {
        mean_o[row] = mean; rstd_o[row] = rstd;
        if (STABLE) inv_amax_o[row] = inv_amax;
    }
    float sat = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        if (e >= D) continue;
        const float4 wv = *reinterpret_cast<const float4*>(w + e);
        const float4 bv = *reinterpret_cast<const float4*>(b + e);
        const float y0 = (xv.v[it].x - mean) * rstd * wv.x + bv.x;
        const float y1 = (xv.v[it].y - mean) * rstd * wv.y + bv.y;
        const float y2 = (xv.v[it].z - mean) * rstd * wv.z + bv.z;
        const float y3 = (xv.v[it].w - mean) * rstd * wv.w + bv.w;
        if (MODE == 0) {
            store_ln_shifted(out_hi, out_lo, row, e, D, shift_ntok, shift_fmap, y0, y1, y2, y3, lo_f16, &sat);
        } else {
            const float4 rv = *reinterpret_cast<const float4*>(resid + row * D + e);
            // lo_f16 & 4 (AMDNUWA_LN_RESID_MINUS): resid - LN(x) -- a reversible block's input from its output, x2 = y2 - g(y1), in one pass
            // (the same bits as -((-y2) + g): the sign flips are exact)
            const float sg = (lo_f16 & 4) ? -1.f : 1.f;
            *reinterpret_cast<float4*>(out_f32 + row * D + e) = make_float4(fmaf(sg, y0, rv.x), fmaf(sg, y1, rv.y), fmaf(sg, y2, rv.z), fmaf(sg, y3, rv.w));
        }
    }
    if (MODE == 0) f16_sat_commit(sat);
}

// Post-norm + residual of block k fused with the pre-norm (and token shift) of block k+1: the new residual stream row is
// still in registers when its second LayerNorm is taken, so the next block's pre-norm kernel (one more full read of the stream)
// disappears.  Both normalisations are computed exactly as the separate kernels compute them.
template <int NV, bool XBF>
__global__ __launch_bounds__(256) void ln_post_pre_kernel(const float* __restrict__ y, const float* __restrict__ resid,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out_f32, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, bf16_t* __restrict__ h_hi,
                                                          bf16_t* __restrict__ h_lo, float* __restrict__ mean2_o,
                                                          float* __restrict__ rstd2_o, long long R, int D, float eps,
                                                          int shift_ntok, int shift_fmap, int lo_f16) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= R) return;
    RowValsT<NV> xv, rres;
    row_load_t<XBF>(y, (size_t)row * D, D, lane, xv, 0.f);
    row_load(resid + row * D, D, lane, rres);                   // both rows in flight together (the loads are branch-free: row_load_t)
    float mean, rstd;
    row_mean_rstd<NV>(xv, D, lane, eps, mean, rstd);
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        if (e >= D) { xv.v[it] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        const float4 wv = *reinterpret_cast<const float4*>(w + e);
        const float4 bv = *reinterpret_cast<const float4*>(b + e);
        const float4 rv = rres.v[it];
        const float y0 = (xv.v[it].x - mean) * rstd * wv.x + bv.x;
        const float y1 = (xv.v[it].y - mean) * rstd * wv.y + bv.y;
        const float y2 = (xv.v[it].z - mean) * rstd * wv.z + bv.z;
        const float y3 = (xv.v[it].w - mean) * rstd * wv.w + bv.w;
        xv.v[it] = make_float4(rv.x + y0, rv.y + y1, rv.z + y2, rv.w + y3);
        *reinterpret_cast<float4*>(out_f32 + row * D + e) = xv.v[it];
    }
    float mean2, rstd2;
    row_mean_rstd<NV>(xv, D, lane, eps, mean2, rstd2);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; mean2_o[row] = mean2; rstd2_o[row] = rstd2; }
    float sat = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        if (e >= D) continue;
        const float4 wv = *reinterpret_cast<const float4*>(w2 + e);
        const float4 bv = *reinterpret_cast<const float4*>(b2 + e);
        store_ln_shifted(h_hi, h_lo, row, e, D, shift_ntok, shift_fmap,
                         (xv.v[it].x - mean2) * rstd2 * wv.x + bv.x, (xv.v[it].y - mean2) * rstd2 * wv.y + bv.y,
                         (xv.v[it].z - mean2) * rstd2 * wv.z + bv.z, (xv.v[it].w - mean2) * rstd2 * wv.w + bv.w, lo_f16, &sat);
    }
    f16_sat_commit(sat);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  dy fp32 (optionally read through the INVERSE token shift), x = LN input.
//   OUT 0: dx -> bf16 hi/lo (post-norm backward; feeds the dgrad / wgrad GEMMs)
//   OUT 1: dx_acc (fp32) += dx (pre-norm backward; accumulates into the residual-stream gradient)
// Writes per-block partial sums [nblk][3][D]: dw, db, and sum(dx) (= bias grad of the Linear that
// produced x, when there is one).  Grid-stride over rows, fixed order => deterministic.
// ---------------------------------------------------------------------------------------------
// fp16 gradients (round 5; scale2 = device {S, 1 / S}, see common.h): IN == 3 reads dy as fp16(S * value) and multiplies by 1 / S;
// out_f16 != 0 (OUT 0) writes dx_hi = fp16(S * dx), saturating, no lo part.
template <int OUT, bool STABLE, int NV, int IN>          // IN: 0 = dy, x fp32; 1 = x is bf16; 2 = dy is bf16; 3 = dy is fp16, scaled
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const float* __restrict__ inv_amax_i, const float* __restrict__ w,
                                                     bf16_t* __restrict__ dx_hi, bf16_t* __restrict__ dx_lo,
                                                     float* dx_acc, const float* dres, float* __restrict__ partial,
                                                     long long R, int D, int shift_ntok, int shift_fmap,
                                                     const float* __restrict__ scale2, int out_f16) {
#pragma clang fp contract(off)        // this kernel and the chained one must round identically: their results are compared bit for bit
    __shared__ float red[ROWS_PER_BLOCK][3][NV * 256];
    const int lane = threadIdx.x & 63, wv_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row math on the SALU
    float4 pw[NV], pb[NV], ps[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) pw[it] = pb[it] = ps[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int quarter = D >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_res = OUT == 1 && dres != nullptr;
    constexpr int DYF = IN == 2 ? 1 : (IN == 3 ? 2 : 0);                 // storage of dy: fp32 / bf16 / fp16
    // (IN == 0 with a scale pair: fp32 dy is multiplied by scale2[1] too -- the upstream gradient of the loss rides into the final norm's
    //  backward as a device scalar instead of costing a read-modify-write pass over dy)
    const bool dy_scaled = IN == 0 && (out_f16 & 2);
    out_f16 &= 1;
    const float gin = (IN == 3 || dy_scaled) ? f16_gs_inv(scale2) : 1.f, gout = out_f16 ? f16_gs(scale2) : 1.f;
    float sat = 0.f;
    struct RowIn { RowValsT<NV> xv, gv, rv; float mean, rstd, ia; };
    // everything one row needs from HBM, issued together (the residual-stream gradient included) ...
    auto load_row = [&](long long row, RowIn& in) {
        row_load_t<IN == 1>(x, (size_t)row * D, D, lane, in.xv);
        if (shift_ntok > 0) {
            // d(unshifted)[i][c] = d(shifted)[i + fmap][c] (quarter 0, when that row took its value from i),
            //                      d(shifted)[i + 1][c]    (quarter 1), d(shifted)[i][c] otherwise
            const int i = (int)(row % shift_ntok);
            long long src_h = -1, src_w = -1;
            const bool audio = shift_fmap < 0;          // ShiftAudioTokens: both channel quarters of the first half come from row i + 1, row 0 included
            if (audio) {
                if (i + 1 < shift_ntok) src_h = src_w = row + 1;
            } else if (i > 0) {
                const int p = i - 1, wq = p % shift_fmap, yq = (p / shift_fmap) % shift_fmap;
                if (yq < shift_fmap - 1 && i + shift_fmap < shift_ntok) src_h = row + shift_fmap;
                if (wq < shift_fmap - 1 && i + 1 < shift_ntok) src_w = row + 1;
            }
#pragma unroll
            for (int it = 0; it < NV; ++it) {                                 // branch-free (see row_load_t)
                const int e0 = (lane + it * 64) * 4;
                const bool inr = e0 < D;
                const int e = inr ? e0 : 0;
                long long src = row;
                if (i > 0 || audio) { if (e < quarter) src = src_h; else if (e < 2 * quarter) src = src_w; }
                const bool has = src >= 0 && inr;
                const float4 v = ld4f<DYF>(dy, (size_t)(has ? src : row) * D + e);
                if (IN == 3 || IN == 0) in.gv.v[it] = make_float4(has ? v.x * gin : 0.f, has ? v.y * gin : 0.f, has ? v.z * gin : 0.f, has ? v.w * gin : 0.f);
                else in.gv.v[it] = make_float4(has ? v.x : 0.f, has ? v.y : 0.f, has ? v.z : 0.f, has ? v.w : 0.f);
            }
        } else {
            if (IN == 3) row_load_f<2>(dy, (size_t)row * D, D, lane, in.gv, gin);
            else if (IN == 0) row_load_f<0>(dy, (size_t)row * D, D, lane, in.gv, gin);
            else row_load_t<IN == 2>(dy, (size_t)row * D, D, lane, in.gv);
        }
        if (OUT == 1) {
            if (has_res) row_load(dres + row * D, D, lane, in.rv);
            else row_load(dx_acc + row * D, D, lane, in.rv);
        }
        in.mean = mean_i[row]; in.rstd = rstd_i[row];
        in.ia = STABLE ? inv_amax_i[row] : 1.f;
    };
    // ... and one row AHEAD of the arithmetic: while row r is reduced and stored, row r + stride is already in flight.
    // Rows are taken grid-stride, so at any moment the chip works on one contiguous window of the tensors.
    const long long stride = (long long)gridDim.x * ROWS_PER_BLOCK;
    long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + wv_;
    // two row buffers used alternately (the loop is unrolled by two): the prefetched row never has to be copied
    RowIn bufA, bufB;
    if (row < R) load_row(row, bufA);
    auto process = [&](const RowIn& cur, long long row) {
        const float mean = cur.mean, rstd = cur.rstd, ia = cur.ia;
        float s1 = 0.f, s2 = 0.f;
        float4 xh[NV], g[NV];
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            const float4 wv = ldp4(w, e, D);             // (lanes past D hold zeros in every row value: their terms vanish)
            float4 xx = cur.xv.v[it];
            const float4 gg = cur.gv.v[it];
            if (STABLE) { xx.x *= ia; xx.y *= ia; xx.z *= ia; xx.w *= ia; }
            xh[it] = make_float4((xx.x - mean) * rstd, (xx.y - mean) * rstd, (xx.z - mean) * rstd, (xx.w - mean) * rstd);
            g[it] = make_float4(gg.x * wv.x, gg.y * wv.y, gg.z * wv.z, gg.w * wv.w);
            s1 += (g[it].x + g[it].y) + (g[it].z + g[it].w);
            s2 += (g[it].x * xh[it].x + g[it].y * xh[it].y) + (g[it].z * xh[it].z + g[it].w * xh[it].w);
            pw[it].x += gg.x * xh[it].x; pw[it].y += gg.y * xh[it].y; pw[it].z += gg.z * xh[it].z; pw[it].w += gg.w * xh[it].w;
            pb[it].x += gg.x; pb[it].y += gg.y; pb[it].z += gg.z; pb[it].w += gg.w;
        }
        const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
        const float sc = STABLE ? rstd * ia : rstd;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            if (e >= D) continue;
            const float d0 = sc * (g[it].x - m1 - xh[it].x * m2), d1 = sc * (g[it].y - m1 - xh[it].y * m2);
            const float d2 = sc * (g[it].z - m1 - xh[it].z * m2), d3 = sc * (g[it].w - m1 - xh[it].w * m2);
            ps[it].x += d0; ps[it].y += d1; ps[it].z += d2; ps[it].w += d3;
            if (OUT == 0) {
                if (out_f16) store_bf16x4(dx_hi + row * D, nullptr, e, d0 * gout, d1 * gout, d2 * gout, d3 * gout, 2, &sat);
                else store_bf16x4(dx_hi + row * D, dx_lo ? dx_lo + row * D : nullptr, e, d0, d1, d2, d3);
            } else {
                const float4 o = cur.rv.v[it];
                *reinterpret_cast<float4*>(dx_acc + row * D + e) = make_float4(o.x + d0, o.y + d1, o.z + d2, o.w + d3);
            }
        }
    };
    for (; row < R; row += 2 * stride) {
        const bool m1 = row + stride < R;
        if (m1) load_row(row + stride, bufB);
        process(bufA, row);
        if (!m1) break;
        if (row + 2 * stride < R) load_row(row + 2 * stride, bufA);
        process(bufB, row + stride);
    }
    if (OUT == 0) f16_sat_commit(sat);
    // block reduce the 4 waves' partials in fixed order
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int e = (lane + it * 64) * 4;
        *reinterpret_cast<float4*>(&red[wv_][0][e]) = pw[it];
        *reinterpret_cast<float4*>(&red[wv_][1][e]) = pb[it];
        *reinterpret_cast<float4*>(&red[wv_][2][e]) = ps[it];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * D; idx += 256) {
        const int k = idx / D, c = idx % D;
        partial[((size_t)blockIdx.x * 3 + k) * D + c] = ((red[0][k][c] + red[1][k][c]) + red[2][k][c]) + red[3][k][c];
    }
}

// ---------------------------------------------------------------------------------------------
// Chained LayerNorm backward across a block boundary: the pre-norm backward of block k+1 produces the residual-stream gradient
// row  dx = g + dLN_pre(dh)  and, while that row is still in registers, the post-norm backward of block k consumes it
// (dy_prev = dLN_post(dx)) -- the second read of dx by a separate kernel disappears.
//   dh (fp32 / bf16, read through the inverse token shift of block k+1), x = stream row (LN input of block k+1), g = gradient of
//   block k+1's output;  y_prev = inner output of block k (post-norm input).
// Partials: partA [nblk][3][D] = (dw_pre, db_pre, -) and partB [nblk][3][D] = (dw_post, db_post, sum(dy_prev)).
// ---------------------------------------------------------------------------------------------
// BFM: storage of the two 16-bit-capable inputs -- 0: dh and y_prev fp32; 1: both bf16; 2: dh bf16, y_prev fp32 (the 'bf16x3-fwd' mode:
// its forward keeps the post-norm input in fp32, its backward produces bf16 dgrad outputs); 3: dh fp16 = fp16(S * value), y_prev fp32
// (the fp16-gradient backward of round 5).  out_f16 != 0: dy_prev leaves as fp16(S * value), saturating.  scale2: device {S, 1 / S}.
template <int NV, int BFM, int NT = 0>          // NT bit 0: non-temporal stores, bit 1: non-temporal loads (tuning key 11)
__global__ __launch_bounds__(256) void ln_bwd_chain_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                           const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                           const float* __restrict__ w, const float* __restrict__ gres,
                                                           float* __restrict__ dx_out, const float* __restrict__ yprev,
                                                           const float* __restrict__ meanp_i, const float* __restrict__ rstdp_i,
                                                           const float* __restrict__ wprev, bf16_t* __restrict__ dyp_hi,
                                                           bf16_t* __restrict__ dyp_lo, float* __restrict__ partA,
                                                           float* __restrict__ partB, long long R, int D, int shift_ntok,
                                                           int shift_fmap, const float* __restrict__ scale2, int out_f16) {
#pragma clang fp contract(off)        // (see ln_bwd_kernel)
    constexpr bool BF = BFM == 1 || BFM == 2, YBF = BFM == 1;
    constexpr int DHF = BFM == 3 ? 2 : (BF ? 1 : 0);                     // storage of dh: fp32 / bf16 / fp16
    const float gin = BFM == 3 ? f16_gs_inv(scale2) : 1.f, gout = out_f16 ? f16_gs(scale2) : 1.f;
    float sat = 0.f;
    __shared__ float red[ROWS_PER_BLOCK][3][NV * 256];
    const int lane = threadIdx.x & 63, wv_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 pwA[NV], pbA[NV], pwB[NV], pbB[NV], psB[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) pwA[it] = pbA[it] = pwB[it] = pbB[it] = psB[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int quarter = D >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    struct RowIn { RowValsT<NV> xv, gv, rv, yv; float mean, rstd, meanp, rstdp; };
    auto load_row = [&](long long row, RowIn& in) {
        if (NT & 2) {
#pragma unroll
            for (int it = 0; it < NV; ++it) { const int e = (lane + it * 64) * 4; in.xv.v[it] = e < D ? ld4_nt<false>(x, (size_t)row * D + e) : zero4; }
        } else row_load(x + row * D, D, lane, in.xv);
        if (shift_ntok > 0) {
            const int i = (int)(row % shift_ntok);
            long long src_h = -1, src_w = -1;
            const bool audio = shift_fmap < 0;          // ShiftAudioTokens: both channel quarters of the first half come from row i + 1, row 0 included
            if (audio) {
                if (i + 1 < shift_ntok) src_h = src_w = row + 1;
            } else if (i > 0) {
                const int p = i - 1, wq = p % shift_fmap, yq = (p / shift_fmap) % shift_fmap;
                if (yq < shift_fmap - 1 && i + shift_fmap < shift_ntok) src_h = row + shift_fmap;
                if (wq < shift_fmap - 1 && i + 1 < shift_ntok) src_w = row + 1;
            }
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                const int e0 = (lane + it * 64) * 4;
                const bool inr = e0 < D;
                const int e = inr ? e0 : 0;
                long long src = row;
                if (i > 0 || audio) { if (e < quarter) src = src_h; else if (e < 2 * quarter) src = src_w; }
                const bool has = src >= 0 && inr;                                    // branch-free: rows without a source re-read their own row
                const size_t so = (size_t)(has ? src : row) * D + e;
                const float4 v = BFM == 3 ? ld4h(dh, so) : ((NT & 2) ? ld4_nt<BF>(dh, so) : ld4<BF>(dh, so));
                in.gv.v[it] = make_float4(has ? v.x : 0.f, has ? v.y : 0.f, has ? v.z : 0.f, has ? v.w : 0.f);
            }
        } else {
            if (BFM == 3) row_load_f<2>(dh, (size_t)row * D, D, lane, in.gv);
            else row_load_t<BF>(dh, (size_t)row * D, D, lane, in.gv);
        }
        if (NT & 2) {
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                const int e = (lane + it * 64) * 4;
                in.rv.v[it] = e < D ? ld4_nt<false>(gres, (size_t)row * D + e) : zero4;
                in.yv.v[it] = e < D ? ld4_nt<YBF>(yprev, (size_t)row * D + e) : zero4;
            }
        } else {
            row_load(gres + row * D, D, lane, in.rv);
            row_load_t<YBF>(yprev, (size_t)row * D, D, lane, in.yv);
        }
        in.mean = mean_i[row]; in.rstd = rstd_i[row]; in.meanp = meanp_i[row]; in.rstdp = rstdp_i[row];
    };
    const long long stride = (long long)gridDim.x * ROWS_PER_BLOCK;
    long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + wv_;
    // two row buffers used alternately (the loop is unrolled by two): the prefetched row never has to be copied
    RowIn bufA, bufB;
    if (row < R) load_row(row, bufA);
    auto process = [&](const RowIn& cur, long long row) {
        // ---- pre-norm backward of block k+1 -> dx
        float s1 = 0.f, s2 = 0.f;
        float4 xh[NV], g[NV];
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            const float4 wv = ldp4(w, e, D);             // (lanes past D hold zeros in every row value: their terms vanish)
            const float4 xx = cur.xv.v[it], gg = cur.gv.v[it];
            xh[it] = make_float4((xx.x - cur.mean) * cur.rstd, (xx.y - cur.mean) * cur.rstd, (xx.z - cur.mean) * cur.rstd, (xx.w - cur.mean) * cur.rstd);
            g[it] = make_float4(gg.x * wv.x, gg.y * wv.y, gg.z * wv.z, gg.w * wv.w);
            s1 += (g[it].x + g[it].y) + (g[it].z + g[it].w);
            s2 += (g[it].x * xh[it].x + g[it].y * xh[it].y) + (g[it].z * xh[it].z + g[it].w * xh[it].w);
            pwA[it].x += gg.x * xh[it].x; pwA[it].y += gg.y * xh[it].y; pwA[it].z += gg.z * xh[it].z; pwA[it].w += gg.w * xh[it].w;
            pbA[it].x += gg.x; pbA[it].y += gg.y; pbA[it].z += gg.z; pbA[it].w += gg.w;
        }
        const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
        float4 dx[NV];
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            if (e >= D) { dx[it] = zero4; continue; }
            const float4 o = cur.rv.v[it];
            // (fp16 dh = S * value: every term in the bracket is linear in dh, so the 1 / S rides on rstd -- one scalar product per row
            //  instead of one per element; the weight-gradient partials pwA / pbA are unscaled once, after the row loop)
            const float rs = BFM == 3 ? cur.rstd * gin : cur.rstd;
            dx[it] = make_float4(o.x + rs * (g[it].x - m1 - xh[it].x * m2), o.y + rs * (g[it].y - m1 - xh[it].y * m2),
                                 o.z + rs * (g[it].z - m1 - xh[it].z * m2), o.w + rs * (g[it].w - m1 - xh[it].w * m2));
            if (NT & 1) st4_nt(dx_out + row * D + e, dx[it]); else *reinterpret_cast<float4*>(dx_out + row * D + e) = dx[it];
        }
        // ---- post-norm backward of block k on the row just produced -> dy_prev
        s1 = 0.f; s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            const float4 wv = ldp4(wprev, e, D);
            const float4 yy = cur.yv.v[it], gg = dx[it];
            xh[it] = make_float4((yy.x - cur.meanp) * cur.rstdp, (yy.y - cur.meanp) * cur.rstdp, (yy.z - cur.meanp) * cur.rstdp, (yy.w - cur.meanp) * cur.rstdp);
            g[it] = make_float4(gg.x * wv.x, gg.y * wv.y, gg.z * wv.z, gg.w * wv.w);
            s1 += (g[it].x + g[it].y) + (g[it].z + g[it].w);
            s2 += (g[it].x * xh[it].x + g[it].y * xh[it].y) + (g[it].z * xh[it].z + g[it].w * xh[it].w);
            pwB[it].x += gg.x * xh[it].x; pwB[it].y += gg.y * xh[it].y; pwB[it].z += gg.z * xh[it].z; pwB[it].w += gg.w * xh[it].w;
            pbB[it].x += gg.x; pbB[it].y += gg.y; pbB[it].z += gg.z; pbB[it].w += gg.w;
        }
        const float n1 = wave_sum(s1) / D, n2 = wave_sum(s2) / D;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            if (e >= D) continue;
            // (fp16 dy_prev = S * value: S rides on rstd_prev; the column sums psB are unscaled once, after the row loop -- S is a power of two)
            const float rp = cur.rstdp * gout;
            const float d0 = rp * (g[it].x - n1 - xh[it].x * n2), d1 = rp * (g[it].y - n1 - xh[it].y * n2);
            const float d2 = rp * (g[it].z - n1 - xh[it].z * n2), d3 = rp * (g[it].w - n1 - xh[it].w * n2);
            psB[it].x += d0; psB[it].y += d1; psB[it].z += d2; psB[it].w += d3;
            if (out_f16) store_bf16x4(dyp_hi + row * D, nullptr, e, d0, d1, d2, d3, 2, &sat);
            else if ((NT & 1) && !dyp_lo) st_bf16x4_nt(dyp_hi + row * D + e, d0, d1, d2, d3);
            else store_bf16x4(dyp_hi + row * D, dyp_lo ? dyp_lo + row * D : nullptr, e, d0, d1, d2, d3);
        }
    };
    for (; row < R; row += 2 * stride) {
        const bool m1 = row + stride < R;
        if (m1) load_row(row + stride, bufB);
        process(bufA, row);
        if (!m1) break;
        if (row + 2 * stride < R) load_row(row + 2 * stride, bufA);
        process(bufB, row + stride);
    }
    f16_sat_commit(sat);
    if (BFM == 3 || out_f16) {
        const float ginv = BFM == 3 ? gin : 1.f, goutinv = out_f16 ? f16_gs_inv(scale2) : 1.f;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            pwA[it].x *= ginv; pwA[it].y *= ginv; pwA[it].z *= ginv; pwA[it].w *= ginv;
            pbA[it].x *= ginv; pbA[it].y *= ginv; pbA[it].z *= ginv; pbA[it].w *= ginv;
            psB[it].x *= goutinv; psB[it].y *= goutinv; psB[it].z *= goutinv; psB[it].w *= goutinv;
        }
    }
    // block reduce the 4 waves' partials in fixed order, one LayerNorm at a time through the same LDS
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int e = (lane + it * 64) * 4;
            *reinterpret_cast<float4*>(&red[wv_][0][e]) = pass ? pwB[it] : pwA[it];
            *reinterpret_cast<float4*>(&red[wv_][1][e]) = pass ? pbB[it] : pbA[it];
            *reinterpret_cast<float4*>(&red[wv_][2][e]) = pass ? psB[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        float* part = pass ? partB : partA;
        for (int idx = threadIdx.x; idx < 3 * D; idx += 256) {
            const int k = idx / D, c = idx % D;
            part[((size_t)blockIdx.x * 3 + k) * D + c] = ((red[0][k][c] + red[1][k][c]) + red[2][k][c]) + red[3][k][c];
        }
    }
}

// column sums of an fp32 [R, D] matrix -> partial[blk][D] (slot k = 0 of a 1-row partial layout)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ partial, long long R, int D) {
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float s = 0.f;
        for (long long r = blockIdx.x; r < R; r += gridDim.x) s += x[r * D + c];
        partial[(size_t)blockIdx.x * D + c] = s;
    }
}

// out[k][c] (+)= sum over blocks of partial[blk][k][c]  for the selected k rows -> destination pointers
// 1024 threads = 64 consecutive (k, c) columns x 16 row groups: coalesced 256-byte reads, 16-way
// row parallelism, fixed combine order (deterministic).  grid = ceil(nk*D / 64).
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const float* __restrict__ partial, int nblk, int nk, int D,
                                                              float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2, int accumulate) {
    // 1024 threads = 16 consecutive (k, c) columns x 64 row groups (64-byte segments per row group; 4x the blocks of a
    // 64-column layout, which left most of the chip idle on these few-MB reductions); fixed combine order.
    __shared__ float red[64][17];
    const int col = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + col;
    float s = 0.f;
    if (idx < nk * D)
        for (int bidx = rg; bidx < nblk; bidx += 64 * 4) {          // four partials in flight, added in index order
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bu = bidx + 64 * u;
                const float t = partial[(size_t)(bu < nblk ? bu : 0) * nk * D + idx];
                v[u] = bu < nblk ? t : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && idx < nk * D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) t += red[r][col];
        const int k = idx / D, c = idx % D;
        float* o = k == 0 ? o0 : (k == 1 ? o1 : o2);
        if (o) o[c] = accumulate ? o[c] + t : t;
    }
}

// ---------------------------------------------------------------------------------------------
// GEGLU (np.py:255-258): u = [a | g] (each FP wide, bf16 hi[/lo]) -> gg = a * gelu(g)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float hl(const bf16_t* hi, const bf16_t* lo, size_t i) { return bf2f(hi[i]) + (lo ? bf2f(lo[i]) : 0.f); }

__global__ void geglu_fwd_generic_kernel(const bf16_t* __restrict__ u_hi, const bf16_t* __restrict__ u_lo,
                                 bf16_t* __restrict__ o_hi, bf16_t* __restrict__ o_lo, long long R, int FP) {
    const size_t total = (size_t)R * FP / 4;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t row = (t * 4) / FP;
        const int c = (int)((t * 4) % FP);
        const size_t ia = row * 2 * FP + c, ig = ia + FP;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = hl(u_hi, u_lo, ia + k) * gelu_f(hl(u_hi, u_lo, ig + k));
        store_bf16x4(o_hi + row * FP, o_lo ? o_lo + row * FP : nullptr, c, o[0], o[1], o[2], o[3]);
    }
}

__global__ void geglu_bwd_generic_kernel(const bf16_t* __restrict__ u_hi, const bf16_t* __restrict__ u_lo,
                                 const bf16_t* __restrict__ d_hi, const bf16_t* __restrict__ d_lo,
                                 bf16_t* __restrict__ du_hi, bf16_t* __restrict__ du_lo, long long R, int FP) {
    const size_t total = (size_t)R * FP / 4;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t row = (t * 4) / FP;
        const int c = (int)((t * 4) % FP);
        const size_t ia = row * 2 * FP + c, ig = ia + FP, id = row * FP + c;
        float da[4], dg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = hl(u_hi, u_lo, ia + k), g = hl(u_hi, u_lo, ig + k), d = hl(d_hi, d_lo, id + k);
            da[k] = d * gelu_f(g);
            dg[k] = d * a * gelu_grad_f(g);
        }
        store_bf16x4(du_hi + row * 2 * FP, du_lo ? du_lo + row * 2 * FP : nullptr, c, da[0], da[1], da[2], da[3]);
        store_bf16x4(du_hi + row * 2 * FP, du_lo ? du_lo + row * 2 * FP : nullptr, c + FP, dg[0], dg[1], dg[2], dg[3]);
    }
}

// row-streaming forms (FP % 8 == 0): one wave per token row, 16-byte accesses, rows visited in launch order like the LN kernels
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(w[k] << 16); f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}
template <bool LO>
__device__ __forceinline__ void load8(const bf16_t* hi, const bf16_t* lo, size_t off, float* f) {
    unpack8(*reinterpret_cast<const uint4*>(hi + off), f);
    if (LO) {
        float l[8];
        unpack8(*reinterpret_cast<const uint4*>(lo + off), l);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += l[k];
    }
}
template <bool LO>
__device__ __forceinline__ void store8(bf16_t* hi, bf16_t* lo, size_t off, const float* f) {
    if (!LO) {
        *reinterpret_cast<uint4*>(hi + off) = make_uint4(pack2_rne(f[0], f[1]), pack2_rne(f[2], f[3]), pack2_rne(f[4], f[5]), pack2_rne(f[6], f[7]));
        return;
    }
    bf16_t h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f2bf_hilo(f[k], h[k], l[k]);
    *reinterpret_cast<uint4*>(hi + off) = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
    if (LO) *reinterpret_cast<uint4*>(lo + off) = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
}

// IL: u in the interleaved-by-8 layout (columns [16 q, 16 q + 8) = a_{8q..8q+7}, [16 q + 8, 16 q + 16) = the matching gates) that
// lets the FF1 GEMM epilogue apply the gate itself: a lane of that GEMM owns 16 contiguous output columns
template <bool LO, bool IL = false>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* __restrict__ u_hi, const bf16_t* __restrict__ u_lo,
                                                        bf16_t* __restrict__ o_hi, bf16_t* __restrict__ o_lo, long long R, int FP) {
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= R) return;
    const size_t ub = (size_t)row * 2 * FP, ob = (size_t)row * FP;
    for (int c = (threadIdx.x & 63) * 8; c < FP; c += 512) {
        float a[8], g[8], o[8];
        load8<LO>(u_hi, u_lo, IL ? ub + 2 * c : ub + c, a);
        load8<LO>(u_hi, u_lo, IL ? ub + 2 * c + 8 : ub + FP + c, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = a[k] * gelu_f(g[k]);
        store8<LO>(o_hi, o_lo, ob + c, o);
    }
}

template <bool LO, bool IL = false>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ u_hi, const bf16_t* __restrict__ u_lo,
                                                        const bf16_t* __restrict__ d_hi, const bf16_t* __restrict__ d_lo,
                                                        bf16_t* __restrict__ du_hi, bf16_t* __restrict__ du_lo, long long R, int FP) {
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= R) return;
    const size_t ub = (size_t)row * 2 * FP, db = (size_t)row * FP;
    for (int c = (threadIdx.x & 63) * 8; c < FP; c += 512) {
        float a[8], g[8], d[8], da[8], dg[8];
        load8<LO>(u_hi, u_lo, IL ? ub + 2 * c : ub + c, a);
        load8<LO>(u_hi, u_lo, IL ? ub + 2 * c + 8 : ub + FP + c, g);
        load8<LO>(d_hi, d_lo, db + c, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float y, dy;
            gelu_both_f(g[k], y, dy);
            da[k] = d[k] * y;
            dg[k] = d[k] * a[k] * dy;
        }
        store8<LO>(du_hi, du_lo, IL ? ub + 2 * c : ub + c, da);
        store8<LO>(du_hi, du_lo, IL ? ub + 2 * c + 8 : ub + FP + c, dg);
    }
}

// ---------------------------------------------------------------------------------------------
// casts / transposes (weight preparation, once per optimiser step; small activations)
// ---------------------------------------------------------------------------------------------
// dst[r][c] (bf16 hi/lo, ld = ldd) = src[r][c] for c < C, 0 for C <= c < Cp
__global__ void cast_pad_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                int ldd, long long R, int C, int Cp) {
    const size_t total = (size_t)R * Cp;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / Cp;
        const int c = (int)(t % Cp);
        const float v = c < C ? src[r * lds_ + c] : 0.f;
        bf16_t h, l;
        f2bf_hilo(v, h, l);
        hi[r * ldd + c] = h;
        if (lo) lo[r * ldd + c] = l;
    }
}

// dst[c][r] (bf16 hi/lo, ld = ldd) = src[r][c]; 32x32 LDS tile transpose
__global__ void transpose_cast_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ hi,
                                      bf16_t* __restrict__ lo, int ldd, int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < C && r < R) {
            bf16_t h, l;
            f2bf_hilo(tile[tx][k], h, l);
            hi[(size_t)c * ldd + r] = h;
            if (lo) lo[(size_t)c * ldd + r] = l;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// embedding assemble (np.py:1940-1944): x[b, 0] = bos; x[b, 1+p] = ((ax1[f] + ax2[y]) + ax3[w]) + frac_grad(W[id])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ W,
                                                        const float* __restrict__ ax1, const float* __restrict__ ax2,
                                                        const float* __restrict__ ax3, const float* __restrict__ bos,
                                                        float* __restrict__ x, int B, int ntok, int D, int H, int Wd, float frac) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= (long long)B * ntok) return;
    const int b = (int)(row / ntok), i = (int)(row % ntok);
    float* o = x + row * D;
    if (i == 0) {
        for (int e = lane * 4; e < D; e += 256) *reinterpret_cast<float4*>(o + e) = *reinterpret_cast<const float4*>(bos + e);
        return;
    }
    const int p = i - 1;
    const int w = p % Wd, y = (p / Wd) % H, f = p / (Wd * H);
    const long long id = ids[(long long)b * (ntok - 1) + p];
    const float* wr = W + id * D;
    for (int e = lane * 4; e < D; e += 256) {
        const float4 t = *reinterpret_cast<const float4*>(wr + e);
        const float4 a1 = *reinterpret_cast<const float4*>(ax1 + (size_t)f * D + e);
        const float4 a2 = *reinterpret_cast<const float4*>(ax2 + (size_t)y * D + e);
        const float4 a3 = *reinterpret_cast<const float4*>(ax3 + (size_t)w * D + e);
        float4 em = t;
        if (frac < 1.f) {   // frac_gradient np.py:83-84: t*frac + t.detach()*(1-frac)
            const float omf = 1.f - frac;
            em = make_float4(t.x * frac + t.x * omf, t.y * frac + t.y * omf, t.z * frac + t.z * omf, t.w * frac + t.w * omf);
        }
        *reinterpret_cast<float4*>(o + e) = make_float4(((a1.x + a2.x) + a3.x) + em.x, ((a1.y + a2.y) + a3.y) + em.y,
                                                        ((a1.z + a2.z) + a3.z) + em.z, ((a1.w + a2.w) + a3.w) + em.w);
    }
}

// token-embedding gradient: dW[id] += frac * dx  (fp32 atomics: summation order is not fixed)
__global__ __launch_bounds__(256) void embed_bwd_tok_kernel(const long long* __restrict__ ids, const float* __restrict__ dx,
                                                            float* __restrict__ dW, int B, int ntok, int D, float frac) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= (long long)B * ntok) return;
    const int b = (int)(row / ntok), i = (int)(row % ntok);
    if (i == 0) return;
    const long long id = ids[(long long)b * (ntok - 1) + (i - 1)];
    for (int e = lane; e < D; e += 64) atomicAdd(dW + id * D + e, frac * dx[row * D + e]);
}

// deterministic form: the caller passes the token ids stably sorted (sid) with their original flat positions (perm); the wave
// sitting on the head of a run of equal ids sums that run's rows in position order and owns dW[id] -- fixed order, no atomics
__global__ __launch_bounds__(256) void embed_bwd_tok_sorted_kernel(const long long* __restrict__ sid, const long long* __restrict__ perm,
                                                                   const float* __restrict__ dx, float* __restrict__ dW, long long N,
                                                                   int ntok, int D, float frac) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (j >= N) return;
    const long long id = sid[j];
    if (j > 0 && sid[j - 1] == id) return;
    float acc[MAXV * 4];
#pragma unroll
    for (int e = 0; e < MAXV * 4; ++e) acc[e] = 0.f;
    for (long long jj = j; jj < N && sid[jj] == id; ++jj) {
        const long long src = perm[jj];
        const long long row = (src / (ntok - 1)) * ntok + (src % (ntok - 1)) + 1;
#pragma unroll
        for (int e = 0; e < MAXV * 4; ++e)
            if (lane + 64 * e < D) acc[e] += dx[row * D + lane + 64 * e];
    }
#pragma unroll
    for (int e = 0; e < MAXV * 4; ++e)
        if (lane + 64 * e < D) dW[id * D + lane + 64 * e] += frac * acc[e];
}

// The same sums with LONG runs cut into segments (round 4).  One wave per run is a serial loop over the run: a frequent code (raw
// frames through the tokenizer put most positions on a few codes; padding ids do the same to a text embedding) turned this kernel
// into one wave walking 10^5 rows -- 205 ms per call inside the whole reference step at b = 64 (profiles/r04_full_step_profile.txt).
// Here a wave owns a fixed SEGMENT of S consecutive sorted positions.  Runs that lie inside the segment are summed and added to dW as
// before.  A run that crosses a segment boundary leaves one partial row per segment it touches (the segment's "head" piece when the
// run came in from the left, its "tail" piece when it starts here and goes on to the right); a second kernel lets the wave of the
// segment where the run STARTS add the pieces in segment order.  Positions are summed in sorted order inside a piece and pieces in
// segment order: fixed order, no atomics, and a run of any length costs O(S) + O(length / S) serial row additions.
//   P    [nseg][2][D] floats: head / tail partial of each segment          meta [nseg][4] ints: head valid, head continues, tail valid, -
__global__ __launch_bounds__(256) void embed_bwd_tok_seg_kernel(const long long* __restrict__ sid, const long long* __restrict__ perm,
                                                                const float* __restrict__ dx, float* __restrict__ dW, float* __restrict__ P,
                                                                int* __restrict__ meta, long long N, int S, int nseg, int ntok, int D, float frac) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= nseg) return;
    const long long j0 = (long long)w * S, j1 = j0 + S < N ? j0 + S : N;
    const bool cont_in = j0 > 0 && sid[j0 - 1] == sid[j0];
    float acc[MAXV * 4];
#pragma unroll
    for (int e = 0; e < MAXV * 4; ++e) acc[e] = 0.f;
    long long cur = sid[j0];
    bool first_run = true, head_valid = false;
    auto put = [&](float* dst, bool add) {           // dst[.] (+)= acc (x frac when it goes to dW)
#pragma unroll
        for (int e = 0; e < MAXV * 4; ++e)
            if (lane + 64 * e < D) dst[lane + 64 * e] = add ? dst[lane + 64 * e] + frac * acc[e] : acc[e];
    };
    for (long long jj = j0; jj < j1; ++jj) {
        const long long id = sid[jj];
        if (id != cur) {                             // the run of `cur` ends inside this segment
            if (first_run && cont_in) { put(P + ((size_t)w * 2 + 0) * D, false); head_valid = true; }
            else put(dW + cur * D, true);
#pragma unroll
            for (int e = 0; e < MAXV * 4; ++e) acc[e] = 0.f;
            cur = id; first_run = false;
        }
        const long long src = perm[jj];
        const long long row = (src / (ntok - 1)) * ntok + (src % (ntok - 1)) + 1;
#pragma unroll
        for (int e = 0; e < MAXV * 4; ++e)
            if (lane + 64 * e < D) acc[e] += dx[row * D + lane + 64 * e];
    }
    const bool cont_out = j1 < N && sid[j1] == cur;
    int head_cont = 0, tail_valid = 0;
    if (first_run && cont_in) { put(P + ((size_t)w * 2 + 0) * D, false); head_valid = true; head_cont = cont_out ? 1 : 0; }   // one run fills the segment
    else if (cont_out) { put(P + ((size_t)w * 2 + 1) * D, false); tail_valid = 1; }
    else put(dW + cur * D, true);
    if (lane == 0) *reinterpret_cast<int4*>(meta + (size_t)w * 4) = make_int4(head_valid ? 1 : 0, head_cont, tail_valid, 0);
}
// the segment where a boundary-crossing run STARTS (it holds the run's tail piece) adds the head pieces of the segments that follow
__global__ __launch_bounds__(256) void embed_bwd_tok_join_kernel(const long long* __restrict__ sid, const float* __restrict__ P,
                                                                 const int* __restrict__ meta, float* __restrict__ dW, long long N, int S,
                                                                 int nseg, int D, float frac) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= nseg || !meta[(size_t)w * 4 + 2]) return;
    const long long jl = (long long)(w + 1) * S - 1;
    const long long id = sid[jl < N ? jl : N - 1];                   // the id of the segment's last run
    float acc[MAXV * 4];
#pragma unroll
    for (int e = 0; e < MAXV * 4; ++e) acc[e] = lane + 64 * e < D ? P[((size_t)w * 2 + 1) * D + lane + 64 * e] : 0.f;
    for (int k = w + 1; k < nseg; ++k) {                              // (segment k's head piece exists: the run continued into it)
#pragma unroll
        for (int e = 0; e < MAXV * 4; ++e)
            if (lane + 64 * e < D) acc[e] += P[((size_t)k * 2 + 0) * D + lane + 64 * e];
        if (!meta[(size_t)k * 4 + 1]) break;
    }
#pragma unroll
    for (int e = 0; e < MAXV * 4; ++e)
        if (lane + 64 * e < D) dW[id * D + lane + 64 * e] += frac * acc[e];
}

// axial / bos gradients, two deterministic stages:
//  A: T[p][c] = sum_b dx[b][1+p][c]  (p < ntok-1);  T[ntok-1][c] = sum_b dx[b][0][c]  (bos)
//  B: ax1[f] = sum_{y,w} T, ax2[y] = sum_{f,w} T, ax3[w] = sum_{f,y} T   (one block per axis entry)
__global__ __launch_bounds__(256) void embed_bwd_possum_kernel(const float* __restrict__ dx, float* __restrict__ T,
                                                               float* __restrict__ dbos, int B, int ntok, int D) {
    const int i = blockIdx.x;            // token row inside the sample, 0 = bos
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dx[((size_t)b * ntok + i) * D + c];
        if (i == 0) dbos[c] += s; else T[(size_t)(i - 1) * D + c] = s;
    }
}
__global__ __launch_bounds__(256) void embed_bwd_axial_kernel(const float* __restrict__ T, float* __restrict__ d1,
                                                              float* __restrict__ d2, float* __restrict__ d3,
                                                              int ntok, int D, int F, int H, int Wd) {
    // grid (axis entry k, 64-column chunk); 256 threads = 64 columns x 4 position groups; fixed combine order
    __shared__ float red[4][64];
    const int k = blockIdx.x, lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    float s = 0.f;
    if (c < D)
        for (int p = pg; p < ntok - 1; p += 4) {
            const int w = p % Wd, y = (p / Wd) % H, f = p / (Wd * H);
            const bool hit = k < F ? (f == k) : (k < F + H ? (y == k - F) : (w == k - F - H));
            if (hit) s += T[(size_t)p * D + c];
        }
    red[pg][lane] = s;
    __syncthreads();
    if (pg == 0 && c < D) {
        const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        if (k < F) d1[(size_t)k * D + c] += t;
        else if (k < F + H) d2[(size_t)(k - F) * D + c] += t;
        else d3[(size_t)(k - F - H) * D + c] += t;
    }
}

// ---------------------------------------------------------------------------------------------
// cross entropy over fp32 logits [R, C] (np.py:1963): row loss = lse - logit[target];
// dlogits (bf16 hi/lo) = (softmax - onehot) * grad_scale   (grad_scale = 1/R for the mean)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ tgt,
                                                     float* __restrict__ row_loss, bf16_t* __restrict__ dl_hi,
                                                     bf16_t* __restrict__ dl_lo, int C, int ldd, float grad_scale) {
    __shared__ float sred[4];
    const long long row = blockIdx.x;
    const float* lr = logits + row * C;
    const int tid = threadIdx.x, lane = tid & 63, wv_ = tid >> 6;
    float m = -3.0e38f;
    for (int c = tid * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(lr + c);
        m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    m = wave_max(m);
    if (lane == 0) sred[wv_] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(lr + c);
        s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
    }
    s = wave_sum(s);
    if (lane == 0) sred[wv_] = s;
    __syncthreads();
    s = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float lse = m + logf(s);
    const long long t = tgt[row];
    const bool t_ok = t >= 0 && t < (long long)C;            // an id outside the vocabulary poisons the loss instead of reading out of bounds
    if (tid == 0) row_loss[row] = t_ok ? lse - lr[t] : __builtin_nanf("");
    if (dl_hi) {
        const float inv = 1.f / s;
        for (int c = tid * 4; c < C; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(lr + c);
            float pz[4] = {expf(v.x - m) * inv, expf(v.y - m) * inv, expf(v.z - m) * inv, expf(v.w - m) * inv};
#pragma unroll
            for (int k = 0; k < 4; ++k) pz[k] = (pz[k] - ((long long)(c + k) == t ? 1.f : 0.f)) * grad_scale;
            store_bf16x4(dl_hi + row * ldd, dl_lo ? dl_lo + row * ldd : nullptr, c, pz[0], pz[1], pz[2], pz[3]);
        }
    }
}

// The same kernel with the row held in REGISTERS (C <= 1024 * NV: 8192 logits = 32 floats per thread): one read of the fp32 logits
// instead of three (the row max, the exponential sum and the gradient pass each re-read 32 KB per row: FETCH_SIZE = 3.0 x the tensor
// in the round-3 counter pass).  Same operations in the same order -> bit-identical loss and dlogits.
template <int NV>
__global__ __launch_bounds__(256) void ce_fwd_reg_kernel(const float* __restrict__ logits, const long long* __restrict__ tgt,
                                                         float* __restrict__ row_loss, bf16_t* __restrict__ dl_hi,
                                                         bf16_t* __restrict__ dl_lo, int C, int ldd, float grad_scale) {
    __shared__ float sred[4];
    const long long row = blockIdx.x;
    const float* lr = logits + row * C;
    const int tid = threadIdx.x, lane = tid & 63, wv_ = tid >> 6;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tid * 4 + i * 1024;
        const float4 ld = *reinterpret_cast<const float4*>(lr + (c < C ? c : 0));       // branch-free: all NV loads in flight together
        v[i] = c < C ? ld : make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    }
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (tid * 4 + i * 1024 < C) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    m = wave_max(m);
    if (lane == 0) sred[wv_] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (tid * 4 + i * 1024 < C) s += (expf(v[i].x - m) + expf(v[i].y - m)) + (expf(v[i].z - m) + expf(v[i].w - m));
    s = wave_sum(s);
    if (lane == 0) sred[wv_] = s;
    __syncthreads();
    s = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float lse = m + logf(s);
    const long long t = tgt[row];
    const bool t_ok = t >= 0 && t < (long long)C;
    if (tid == 0) row_loss[row] = t_ok ? lse - lr[t] : __builtin_nanf("");
    if (dl_hi) {
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tid * 4 + i * 1024;
            if (c >= C) continue;
            float pz[4] = {expf(v[i].x - m) * inv, expf(v[i].y - m) * inv, expf(v[i].z - m) * inv, expf(v[i].w - m) * inv};
#pragma unroll
            for (int k = 0; k < 4; ++k) pz[k] = (pz[k] - ((long long)(c + k) == t ? 1.f : 0.f)) * grad_scale;
            store_bf16x4(dl_hi + row * ldd, dl_lo ? dl_lo + row * ldd : nullptr, c, pz[0], pz[1], pz[2], pz[3]);
        }
    }
}

// loss = mean(row_loss) in a fixed order (single block of 1024 threads; a thread keeps four independent float4 loads in flight -- the
// first form, 256 threads adding one dependent 4-byte load after another, took 0.53 ms per step for 1.3 MB)
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ v, long long n, float* __restrict__ out) {
    __shared__ float sred[1024];
    float s = 0.f;
    const long long n4 = (reinterpret_cast<size_t>(v) & 15) == 0 ? n / 4 : 0;
    const float4* v4 = reinterpret_cast<const float4*>(v);
    long long i = threadIdx.x;
    for (; i + 3 * 1024 < n4; i += 4 * 1024) {
        const float4 a = v4[i], b = v4[i + 1024], c = v4[i + 2048], d = v4[i + 3072];
        s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
        s += ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
    }
    for (; i < n4; i += 1024) { const float4 a = v4[i]; s += (a.x + a.y) + (a.z + a.w); }
    for (long long k = n4 * 4 + threadIdx.x; k < n; k += 1024) s += v[k];
    sred[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sred[0] / (float)n;
}

// x[i] *= *scalar  (device scalar: upstream gradient of the loss)
__global__ void scale_by_dev_scalar_kernel(float* __restrict__ x, size_t n, const float* __restrict__ s) {
    const float sc = *s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= sc;
}

inline int grid_for(size_t work, int per_block = 256, int cap = 4096) {
    size_t g = (work + per_block - 1) / per_block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int amdnuwa_ln_fwd(const float* x, const float* resid, const float* w, const float* b, uint16_t* out_hi,
                              uint16_t* out_lo, float* out_f32, float* mean, float* rstd, float* inv_amax, long long R,
                              int D, int mode, int stable, float eps, int shift_ntok, int shift_fmap, hipStream_t stream) {
    const bool xbf = (mode & AMDNUWA_LN_X_BF16) != 0;        // x points at bf16 values
    int lo_f16 = (mode & AMDNUWA_LN_OUT_F16) ? 2 : ((mode & AMDNUWA_LN_LO_F16) ? 1 : 0);   // 1: out_lo = fp16 copy; 2: out_hi itself is fp16
    if (lo_f16 == 2 && out_lo) return AMDNUWA_ERR_ARG;
    if (mode & AMDNUWA_LN_RESID_MINUS) { if (!(mode & 1)) return AMDNUWA_ERR_ARG; lo_f16 = 4; }     // (mode 1 only: out_f32 = resid - LN(x))
    mode &= 1;
    if (shift_ntok > 0 && (mode != 0 || shift_fmap == 0 || shift_fmap < -1 || D % 16)) return AMDNUWA_ERR_ARG;
    if (!x || !w || !b || !mean || !rstd || D % 4 || D > MAXV * 256 || D <= 0) return AMDNUWA_ERR_ARG;
    if (mode == 0 && !out_hi) return AMDNUWA_ERR_ARG;
    if (mode == 1 && (!out_f32 || !resid || stable)) return AMDNUWA_ERR_ARG;
    if (stable && !inv_amax) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    dim3 grid((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
#define LNF(MO, ST, NV_) do { if (xbf) hipLaunchKernelGGL((ln_fwd_kernel<MO, ST, NV_, true>), grid, block, 0, stream, x, resid, w, b, out_hi, out_lo, out_f32, mean, rstd, inv_amax, R, D, eps, shift_ntok, shift_fmap, lo_f16); \
                              else hipLaunchKernelGGL((ln_fwd_kernel<MO, ST, NV_, false>), grid, block, 0, stream, x, resid, w, b, out_hi, out_lo, out_f32, mean, rstd, inv_amax, R, D, eps, shift_ntok, shift_fmap, lo_f16); } while (0)
#define LNF_NV(MO, ST) do { if (D <= 256) LNF(MO, ST, 1); else if (D <= 512) LNF(MO, ST, 2); else LNF(MO, ST, 4); } while (0)
    if (mode == 0 && !stable) LNF_NV(0, false);
    else if (mode == 0) LNF_NV(0, true);
    else LNF_NV(1, false);
#undef LNF_NV
#undef LNF
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_ln_post_pre_fwd(const float* y, const float* resid, const float* w, const float* b, float* out_f32,
                                       float* mean, float* rstd, const float* next_w, const float* next_b, uint16_t* h_hi,
                                       uint16_t* h_lo, float* next_mean, float* next_rstd, long long R, int D, int flags,
                                       float eps, int shift_ntok, int shift_fmap, hipStream_t stream) {
    const bool xbf = (flags & AMDNUWA_LN_X_BF16) != 0;
    const int lo_f16 = (flags & AMDNUWA_LN_OUT_F16) ? 2 : ((flags & AMDNUWA_LN_LO_F16) ? 1 : 0);
    if (lo_f16 == 2 && h_lo) return AMDNUWA_ERR_ARG;
    if (!y || !resid || !w || !b || !out_f32 || !mean || !rstd || !next_w || !next_b || !h_hi || !next_mean || !next_rstd)
        return AMDNUWA_ERR_ARG;
    if (D % 4 || D > MAXV * 256 || D <= 0) return AMDNUWA_ERR_ARG;
    if (shift_ntok > 0 && (shift_fmap == 0 || shift_fmap < -1 || D % 16)) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    dim3 grid((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
#define LPP(NV_) do { if (xbf) hipLaunchKernelGGL((ln_post_pre_kernel<NV_, true>), grid, block, 0, stream, y, resid, w, b, out_f32, mean, rstd, next_w, next_b, h_hi, h_lo, next_mean, next_rstd, R, D, eps, shift_ntok, shift_fmap, lo_f16); \
                      else hipLaunchKernelGGL((ln_post_pre_kernel<NV_, false>), grid, block, 0, stream, y, resid, w, b, out_f32, mean, rstd, next_w, next_b, h_hi, h_lo, next_mean, next_rstd, R, D, eps, shift_ntok, shift_fmap, lo_f16); } while (0)
    if (D <= 256) LPP(1); else if (D <= 512) LPP(2); else LPP(4);
#undef LPP
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

static int ln_bwd_blocks(long long R) {
    long long nb = (R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (nb > 1536) nb = 1536;
    return (int)(nb < 1 ? 1 : nb);
}

extern "C" size_t amdnuwa_ln_bwd_workspace_bytes(long long R, int D) { return (size_t)ln_bwd_blocks(R) * 3 * D * sizeof(float); }

extern "C" int amdnuwa_ln_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* inv_amax,
                              const float* w, uint16_t* dx_hi, uint16_t* dx_lo, float* dx_acc, const float* dres, float* dw, float* db,
                              float* dsum, long long R, int D, int shift_ntok, int shift_fmap, int stable, int accumulate,
                              void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return amdnuwa_ln_bwd_f16(dy, x, mean, rstd, inv_amax, w, dx_hi, dx_lo, dx_acc, dres, dw, db, dsum, R, D, shift_ntok, shift_fmap, stable,
                              accumulate, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int amdnuwa_ln_bwd_f16(const float* dy, const float* x, const float* mean, const float* rstd, const float* inv_amax,
                                  const float* w, uint16_t* dx_hi, uint16_t* dx_lo, float* dx_acc, const float* dres, float* dw, float* db,
                                  float* dsum, long long R, int D, int shift_ntok, int shift_fmap, int stable, int accumulate,
                                  const float* scale2, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const int dy16 = (stable & AMDNUWA_LN_DY_F16) ? 1 : 0;
    const int out_f16 = ((stable & AMDNUWA_LN_OUT_F16) ? 1 : 0) | ((stable & AMDNUWA_LN_DY_SCALED) ? 2 : 0);      // (kernel flag word: bit 0 fp16 output, bit 1 scaled fp32 dy)
    if ((stable & AMDNUWA_LN_DY_SCALED) && ((stable & (AMDNUWA_LN_X_BF16 | AMDNUWA_LN_DY_BF16 | AMDNUWA_LN_DY_F16 | AMDNUWA_LN_OUT_F16)) || !scale2)) return AMDNUWA_ERR_ARG;
    const int in_kind = (stable & AMDNUWA_LN_X_BF16) ? 1 : ((stable & AMDNUWA_LN_DY_BF16) ? 2 : (dy16 ? 3 : 0));
    if ((stable & AMDNUWA_LN_X_BF16) && (stable & (AMDNUWA_LN_DY_BF16 | AMDNUWA_LN_DY_F16))) return AMDNUWA_ERR_UNSUPPORTED;
    if ((stable & AMDNUWA_LN_DY_BF16) && dy16) return AMDNUWA_ERR_ARG;
    if ((out_f16 & 1) && (!dx_hi || dx_lo)) return AMDNUWA_ERR_ARG;
    stable &= 1;
    if (!dy || !x || !mean || !rstd || !w || D % 4 || D > MAXV * 256 || D <= 0) return AMDNUWA_ERR_ARG;
    if ((dx_hi == nullptr) == (dx_acc == nullptr)) return AMDNUWA_ERR_ARG;   // exactly one output form
    if (shift_ntok > 0 && (shift_fmap == 0 || shift_fmap < -1 || D % 16)) return AMDNUWA_ERR_ARG;
    if (stable && !inv_amax) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_ln_bwd_workspace_bytes(R, D)) return AMDNUWA_ERR_WORKSPACE;
    if (R <= 0) return AMDNUWA_OK;
    // one resident wave of blocks exactly (occupancy x CUs): a grid-stride kernel with 1.5 rounds of blocks idles half the chip
    // for its second round
    int nb = ln_bwd_blocks(R);
    float* part = (float*)workspace;
    const dim3 block(256);
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
#define LNB_OCC(OU, ST, NV_, IN_) do { int o_ = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_, ln_bwd_kernel<OU, ST, NV_, IN_>, 256, 0) == hipSuccess && o_ > 0 && o_ * n_cu < nb) nb = o_ * n_cu; } while (0)
#define LNB_(OU, ST, NV_, IN_) do { LNB_OCC(OU, ST, NV_, IN_); hipLaunchKernelGGL((ln_bwd_kernel<OU, ST, NV_, IN_>), dim3(nb), block, 0, stream, dy, x, mean, rstd, inv_amax, w, dx_hi, dx_lo, dx_acc, dres, part, R, D, shift_ntok, shift_fmap, scale2, out_f16); } while (0)
#define LNB(OU, ST, NV_) do { if (in_kind == 1) LNB_(OU, ST, NV_, 1); else if (in_kind == 2) LNB_(OU, ST, NV_, 2); else if (in_kind == 3) LNB_(OU, ST, NV_, 3); else LNB_(OU, ST, NV_, 0); } while (0)
#define LNB_NV(OU, ST) do { if (D <= 256) LNB(OU, ST, 1); else if (D <= 512) LNB(OU, ST, 2); else LNB(OU, ST, 4); } while (0)
    if (dx_hi) { if (stable) LNB_NV(0, true); else LNB_NV(0, false); }
    else       { if (stable) LNB_NV(1, true); else LNB_NV(1, false); }
#undef LNB_NV
#undef LNB
#undef LNB_
#undef LNB_OCC
    LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((3 * D + 15) / 16), dim3(1024), 0, stream, part, nb, 3, D, dw, db, dsum, accumulate);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_ln_bwd_chain_workspace_bytes(long long R, int D) { return 2 * amdnuwa_ln_bwd_workspace_bytes(R, D); }

extern "C" int amdnuwa_ln_bwd_chain(const void* dh, const float* x, const float* mean, const float* rstd, const float* w,
                                    const float* g, float* dx, float* dw, float* db, const void* y_prev, const float* mean_prev,
                                    const float* rstd_prev, const float* w_prev, uint16_t* dy_prev_hi, uint16_t* dy_prev_lo,
                                    float* dw_prev, float* db_prev, float* dsum_prev, long long R, int D, int shift_ntok,
                                    int shift_fmap, int inputs_bf16, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (inputs_bf16 < 0 || inputs_bf16 > 2) return AMDNUWA_ERR_ARG;
    return amdnuwa_ln_bwd_chain_f16(dh, x, mean, rstd, w, g, dx, dw, db, y_prev, mean_prev, rstd_prev, w_prev, dy_prev_hi, dy_prev_lo, dw_prev,
                                    db_prev, dsum_prev, R, D, shift_ntok, shift_fmap, inputs_bf16, 0, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int amdnuwa_ln_bwd_chain_f16(const void* dh, const float* x, const float* mean, const float* rstd, const float* w,
                                        const float* g, float* dx, float* dw, float* db, const void* y_prev, const float* mean_prev,
                                        const float* rstd_prev, const float* w_prev, uint16_t* dy_prev_hi, uint16_t* dy_prev_lo,
                                        float* dw_prev, float* db_prev, float* dsum_prev, long long R, int D, int shift_ntok,
                                        int shift_fmap, int inputs_bf16, int dy_prev_f16, const float* scale2, void* workspace,
                                        size_t workspace_bytes, hipStream_t stream) {
    const int out_f16 = dy_prev_f16 ? 1 : 0;
    if (out_f16 && dy_prev_lo) return AMDNUWA_ERR_ARG;
    if (!dh || !x || !mean || !rstd || !w || !g || !dx || !y_prev || !mean_prev || !rstd_prev || !w_prev || !dy_prev_hi)
        return AMDNUWA_ERR_ARG;
    if (D % 4 || D > MAXV * 256 || D <= 0) return AMDNUWA_ERR_ARG;
    if (shift_ntok > 0 && (shift_fmap == 0 || shift_fmap < -1 || D % 16)) return AMDNUWA_ERR_ARG;
    if (inputs_bf16 < 0 || inputs_bf16 > 3) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_ln_bwd_chain_workspace_bytes(R, D)) return AMDNUWA_ERR_WORKSPACE;
    if (R <= 0) return AMDNUWA_OK;
    int nb = ln_bwd_blocks(R);
    float* partA = (float*)workspace;
    float* partB = partA + (size_t)ln_bwd_blocks(R) * 3 * D;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    const int nt = g_amdnuwa_tuning[11] & 3;
#define LBC__(NV_, BF_, NT_) do { int o_ = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_, ln_bwd_chain_kernel<NV_, BF_, NT_>, 256, 0) == hipSuccess && o_ > 0 && o_ * n_cu < nb) nb = o_ * n_cu; \
        if (g_amdnuwa_tuning[12] > 0 && g_amdnuwa_tuning[12] * n_cu < nb) nb = g_amdnuwa_tuning[12] * n_cu; \
        hipLaunchKernelGGL((ln_bwd_chain_kernel<NV_, BF_, NT_>), dim3(nb), dim3(256), 0, stream, (const float*)dh, x, mean, rstd, w, g, dx, (const float*)y_prev, mean_prev, rstd_prev, w_prev, dy_prev_hi, dy_prev_lo, partA, partB, R, D, shift_ntok, shift_fmap, scale2, out_f16); } while (0)
#define LBC_(NV_, BF_) do { if (nt == 0) LBC__(NV_, BF_, 0); else if (nt == 1) LBC__(NV_, BF_, 1); else if (nt == 2) LBC__(NV_, BF_, 2); else LBC__(NV_, BF_, 3); } while (0)
#define LBC(NV_) do { if (inputs_bf16 == 1) LBC_(NV_, 1); else if (inputs_bf16 == 2) LBC_(NV_, 2); else if (inputs_bf16 == 3) LBC__(NV_, 3, 0); else LBC_(NV_, 0); } while (0)
    if (D <= 256) LBC(1); else if (D <= 512) LBC(2); else LBC(4);
#undef LBC
#undef LBC_
#undef LBC__
    LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((2 * D + 15) / 16), dim3(1024), 0, stream, partA, nb, 3, D, dw, db, (float*)nullptr, 0);
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((3 * D + 15) / 16), dim3(1024), 0, stream, partB, nb, 3, D, dw_prev, db_prev, dsum_prev, 0);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

AMDNUWA_SAT_ACCESSOR(elementwise)

namespace {
__global__ __launch_bounds__(256) void hilo_to_f16_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int ld_in,
                                                          uint16_t* __restrict__ out, int ld_out, long long R, int C) {
    const long long chunks = R * (C / 8);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < chunks; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / (C / 8);
        const int c = (int)(e % (C / 8)) * 8;
        const uint4 a = *reinterpret_cast<const uint4*>(hi + r * ld_in + c);
        const uint4 b = lo ? *reinterpret_cast<const uint4*>(lo + r * ld_in + c) : make_uint4(0, 0, 0, 0);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2_f16_sat(lo_f(aw[k]) + lo_f(bw[k]), hi_f(aw[k]) + hi_f(bw[k]));
        *reinterpret_cast<uint4*>(out + r * ld_out + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
}  // namespace

extern "C" int amdnuwa_hilo_to_f16(const uint16_t* hi, const uint16_t* lo, int ld_in, uint16_t* out, int ld_out, long long R, int C,
                                   hipStream_t stream) {
    if (!hi || !out || C <= 0 || C % 8 || ld_in % 8 || ld_out % 8) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(hilo_to_f16_kernel, dim3(grid_for((size_t)R * (C / 8))), dim3(256), 0, stream, hi, lo, ld_in, out, ld_out, R, C);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_colsum_workspace_bytes(long long R, int D) { return (size_t)(R < 256 ? (R < 1 ? 1 : R) : 256) * D * sizeof(float); }

// out[c] (+)= sum_r x[r][c]   (fixed order => deterministic)
extern "C" int amdnuwa_colsum(const float* x, float* out, long long R, int D, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !out || D <= 0) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_colsum_workspace_bytes(R, D)) return AMDNUWA_ERR_WORKSPACE;
    if (R <= 0) return AMDNUWA_OK;
    const int nb = (int)(R < 256 ? R : 256);
    hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, stream, x, (float*)workspace, R, D);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((D + 15) / 16), dim3(1024), 0, stream, (const float*)workspace, nb, 1, D, out, (float*)nullptr, (float*)nullptr, accumulate);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_geglu_fwd(const uint16_t* u_hi, const uint16_t* u_lo, uint16_t* o_hi, uint16_t* o_lo, long long R, int FP, hipStream_t stream) {
    if (!u_hi || !o_hi || FP % 4) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    const dim3 rg((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    if (FP % 8) hipLaunchKernelGGL(geglu_fwd_generic_kernel, dim3(grid_for((size_t)R * FP / 4)), dim3(256), 0, stream, u_hi, u_lo, o_hi, o_lo, R, FP);
    else if (u_lo && o_lo) hipLaunchKernelGGL((geglu_fwd_kernel<true>), rg, dim3(256), 0, stream, u_hi, u_lo, o_hi, o_lo, R, FP);
    else if (!u_lo && !o_lo) hipLaunchKernelGGL((geglu_fwd_kernel<false>), rg, dim3(256), 0, stream, u_hi, u_lo, o_hi, o_lo, R, FP);
    else hipLaunchKernelGGL(geglu_fwd_generic_kernel, dim3(grid_for((size_t)R * FP / 4)), dim3(256), 0, stream, u_hi, u_lo, o_hi, o_lo, R, FP);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_geglu_il_fwd(const uint16_t* u_hi, const uint16_t* u_lo, uint16_t* o_hi, uint16_t* o_lo, long long R, int FP, hipStream_t stream) {
    if (!u_hi || !o_hi || FP <= 0 || FP % 8 || (u_lo != nullptr) != (o_lo != nullptr)) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    const dim3 rg((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    if (u_lo) hipLaunchKernelGGL((geglu_fwd_kernel<true, true>), rg, dim3(256), 0, stream, u_hi, u_lo, o_hi, o_lo, R, FP);
    else hipLaunchKernelGGL((geglu_fwd_kernel<false, true>), rg, dim3(256), 0, stream, u_hi, u_lo, o_hi, o_lo, R, FP);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_geglu_il_bwd(const uint16_t* u_hi, const uint16_t* u_lo, const uint16_t* d_hi, const uint16_t* d_lo,
                                    uint16_t* du_hi, uint16_t* du_lo, long long R, int FP, hipStream_t stream) {
    if (!u_hi || !d_hi || !du_hi || FP <= 0 || FP % 8) return AMDNUWA_ERR_ARG;
    if ((u_lo != nullptr) != (d_lo != nullptr) || (u_lo != nullptr) != (du_lo != nullptr)) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    const dim3 rg((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    if (u_lo) hipLaunchKernelGGL((geglu_bwd_kernel<true, true>), rg, dim3(256), 0, stream, u_hi, u_lo, d_hi, d_lo, du_hi, du_lo, R, FP);
    else hipLaunchKernelGGL((geglu_bwd_kernel<false, true>), rg, dim3(256), 0, stream, u_hi, u_lo, d_hi, d_lo, du_hi, du_lo, R, FP);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_geglu_bwd(const uint16_t* u_hi, const uint16_t* u_lo, const uint16_t* d_hi, const uint16_t* d_lo,
                                 uint16_t* du_hi, uint16_t* du_lo, long long R, int FP, hipStream_t stream) {
    if (!u_hi || !d_hi || !du_hi || FP % 4) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    const dim3 rg((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    if (FP % 8) hipLaunchKernelGGL(geglu_bwd_generic_kernel, dim3(grid_for((size_t)R * FP / 4)), dim3(256), 0, stream, u_hi, u_lo, d_hi, d_lo, du_hi, du_lo, R, FP);
    else if (u_lo && d_lo && du_lo) hipLaunchKernelGGL((geglu_bwd_kernel<true>), rg, dim3(256), 0, stream, u_hi, u_lo, d_hi, d_lo, du_hi, du_lo, R, FP);
    else if (!u_lo && !d_lo && !du_lo) hipLaunchKernelGGL((geglu_bwd_kernel<false>), rg, dim3(256), 0, stream, u_hi, u_lo, d_hi, d_lo, du_hi, du_lo, R, FP);
    else hipLaunchKernelGGL(geglu_bwd_generic_kernel, dim3(grid_for((size_t)R * FP / 4)), dim3(256), 0, stream, u_hi, u_lo, d_hi, d_lo, du_hi, du_lo, R, FP);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_cast_pad(const float* src, int ld_src, uint16_t* hi, uint16_t* lo, int ld_dst, long long R, int C, int Cp, hipStream_t stream) {
    if (!src || !hi || Cp < C || ld_dst < Cp) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for((size_t)R * Cp)), dim3(256), 0, stream, src, ld_src, hi, lo, ld_dst, R, C, Cp);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_transpose_cast(const float* src, int ld_src, uint16_t* hi, uint16_t* lo, int ld_dst, int R, int C, hipStream_t stream) {
    if (!src || !hi || ld_dst < R) return AMDNUWA_ERR_ARG;
    if (R <= 0 || C <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(transpose_cast_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, src, ld_src, hi, lo, ld_dst, R, C);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_embed_fwd(const long long* ids, const float* W, const float* ax1, const float* ax2, const float* ax3,
                                 const float* bos, float* x, int B, int ntok, int D, int H, int Wd, float frac, hipStream_t stream) {
    if (!ids || !W || !ax1 || !ax2 || !ax3 || !bos || !x || D % 4) return AMDNUWA_ERR_ARG;
    const long long R = (long long)B * ntok;
    if (R <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, stream, ids, W, ax1, ax2, ax3, bos, x, B, ntok, D, H, Wd, frac);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_embed_bwd_workspace_bytes(int ntok, int D) { return (size_t)ntok * D * sizeof(float); }

// all gradient outputs are ACCUMULATED into (caller zero-fills fresh buffers)
extern "C" int amdnuwa_embed_bwd(const long long* ids, const long long* sorted_ids, const long long* perm, const float* dx, float* dW,
                                 float* dax1, float* dax2, float* dax3, float* dbos, int B, int ntok, int D, int F, int H, int Wd,
                                 float frac, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!ids || !dx || !dW || !dax1 || !dax2 || !dax3 || !dbos) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_embed_bwd_workspace_bytes(ntok, D)) return AMDNUWA_ERR_WORKSPACE;
    const long long R = (long long)B * ntok;
    if (R <= 0) return AMDNUWA_OK;
    if (sorted_ids && perm) {
        const long long N = (long long)B * (ntok - 1);
        // segmented form when the workspace (ntok rows of D floats, shared with the positional sums below: they run afterwards on the same
        // stream) holds two partial rows + 4 ints per segment for at least 8 segments; else one wave per run
        const long long rows_ws = (long long)(workspace_bytes / ((size_t)D * sizeof(float)));
        long long nseg_max = (rows_ws * D) / (2LL * D + 4);
        if (N > 0 && nseg_max >= 8 && N >= 64) {
            int S = (int)((N + nseg_max - 1) / nseg_max);
            if (S < 32) S = 32;
            const int nseg = (int)((N + S - 1) / S);
            float* P = (float*)workspace;
            int* meta = (int*)(P + (size_t)nseg * 2 * D);
            hipLaunchKernelGGL(embed_bwd_tok_seg_kernel, dim3((unsigned)((nseg + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, stream, sorted_ids, perm, dx, dW, P, meta, N, S, nseg, ntok, D, frac);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(embed_bwd_tok_join_kernel, dim3((unsigned)((nseg + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, stream, sorted_ids, P, meta, dW, N, S, nseg, D, frac);
        } else if (N > 0)
            hipLaunchKernelGGL(embed_bwd_tok_sorted_kernel, dim3((unsigned)((N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, stream, sorted_ids, perm, dx, dW, N, ntok, D, frac);
    } else {
        hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, stream, ids, dx, dW, B, ntok, D, frac);
    }
    LAUNCH_CHECK();
    float* T = (float*)workspace;
    hipLaunchKernelGGL(embed_bwd_possum_kernel, dim3(ntok), dim3(256), 0, stream, dx, T, dbos, B, ntok, D);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(embed_bwd_axial_kernel, dim3(F + H + Wd, (D + 63) / 64), dim3(256), 0, stream, T, dax1, dax2, dax3, ntok, D, F, H, Wd);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_ce_fwd(const float* logits, const long long* targets, float* row_loss, float* loss, uint16_t* dl_hi,
                              uint16_t* dl_lo, long long R, int C, int ld_dl, float grad_scale, hipStream_t stream) {
    if (!logits || !targets || !row_loss || !loss || C % 4) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    if (C <= 2048)
        hipLaunchKernelGGL(ce_fwd_reg_kernel<2>, dim3((unsigned)R), dim3(256), 0, stream, logits, targets, row_loss, dl_hi, dl_lo, C, ld_dl, grad_scale);
    else if (C <= 8192)
        hipLaunchKernelGGL(ce_fwd_reg_kernel<8>, dim3((unsigned)R), dim3(256), 0, stream, logits, targets, row_loss, dl_hi, dl_lo, C, ld_dl, grad_scale);
    else
        hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, targets, row_loss, dl_hi, dl_lo, C, ld_dl, grad_scale);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, stream, row_loss, R, loss);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_scale_by_device_scalar(float* x, size_t n, const float* scalar, hipStream_t stream) {
    if (!x || !scalar) return AMDNUWA_ERR_ARG;
    if (n == 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(scale_by_dev_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, n, scalar);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
