// Incremental (one new token per sample) decoding kernels for NUWA.generate (np.py:1841-1915) with a key/value cache.
//
// The reference recomputes the whole prefix twice per sampled token.  Every decoder stage is causal -- Sparse3DNA only looks
// at taps <= the query position (np.py:420-457), ShiftVideoTokens only at the row above / to the left (np.py:210-253), and
// cross-attention / FeedForward are row-wise -- so row `pos` of the decoder can be computed from cached rows < pos:
//   * decode_shift_kernel : caches the new pre-norm row and emits its token-shifted form
//   * s3_decode_kernel    : caches the new key / value row and runs the 3DNA attention of the single new query
// The position is read from DEVICE memory so that one captured HIP graph serves every token of the sequence.
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

__device__ __forceinline__ float ld_hl(const bf16_t* hi, const bf16_t* lo, size_t i) {
    return lo ? bf2f(hi[i]) + bf2f(lo[i]) : bf2f(hi[i]);
}
__device__ __forceinline__ void st_hl(bf16_t* hi, bf16_t* lo, size_t i, float v) {
    if (lo) { bf16_t h, l; f2bf_hilo(v, h, l); hi[i] = h; lo[i] = l; }
    else hi[i] = f2bf(v);
}

// h_new [B, D] (row `pos` of every sample) -> cache[b][pos] and out[b] = shift(h)[pos]
__global__ __launch_bounds__(256) void decode_shift_kernel(const bf16_t* __restrict__ h_hi, const bf16_t* __restrict__ h_lo,
                                                           bf16_t* __restrict__ c_hi, bf16_t* __restrict__ c_lo,
                                                           bf16_t* __restrict__ o_hi, bf16_t* __restrict__ o_lo,
                                                           const int* __restrict__ pos_p, int cache_rows, int D, int fmap) {
    const int b = blockIdx.x, pos = pos_p[0];
    if (pos < 0 || pos >= cache_rows) return;
    const size_t crow = ((size_t)b * cache_rows + pos) * D, nrow = (size_t)b * D;
    int yq = 0, wq = 0;
    if (pos > 0) { const int p = pos - 1; wq = p % fmap; yq = (p / fmap) % fmap; }
    const int qd = D >> 2;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const bf16_t hv = h_hi[nrow + c], lv = h_lo ? h_lo[nrow + c] : (bf16_t)0;
        c_hi[crow + c] = hv;
        if (c_lo) c_lo[crow + c] = lv;
        bf16_t oh = hv, ol = lv;
        if (pos > 0 && c < 2 * qd) {
            // first quarter <- the token one grid row up, second quarter <- the token to the left; zero at the frame border
            const bool up = c < qd;
            const bool has = up ? (yq > 0) : (wq > 0);
            const size_t srow = crow - (size_t)(up ? fmap : 1) * D;
            oh = has ? c_hi[srow + c] : (bf16_t)0;
            ol = (has && c_lo) ? c_lo[srow + c] : (bf16_t)0;
        }
        o_hi[nrow + c] = oh;
        if (o_lo) o_lo[nrow + c] = ol;
    }
}

// ---- the norms around a block, for the one new row of every sample, in ONE launch ------------------------------------------
//   x_new = resid + LN(y; w, b)                     (post-norm + residual of the block that just ran; skipped when resid == NULL:
//                                                    then x_new = y, the fp32 decoder input row)
//   h     = LN(x_new; next_w, next_b)               (pre-norm of the next block)
//   cache[b][pos] = h ; out = shift(h)[pos]         (when the next block is token-shifted: cache != NULL)
// One workgroup per sample; D <= 4096.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

template <bool YBF>
__global__ __launch_bounds__(256) void decode_ln_kernel(const void* __restrict__ y_, const float* __restrict__ resid,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ x_new, bf16_t* __restrict__ c_hi,
                                                        bf16_t* __restrict__ c_lo, bf16_t* __restrict__ o_hi,
                                                        bf16_t* __restrict__ o_lo, const int* __restrict__ pos_p, int cache_rows,
                                                        int D, int fmap, float eps) {
    __shared__ float red[8];
    constexpr int MAXI = 4;                         // 4 x 256 threads x 4 elements = 4096
    const int bidx = blockIdx.x, tid = threadIdx.x;
    const size_t row = (size_t)bidx * D;
    float4 v[MAXI];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int e = (tid + it * 256) * 4;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < D) {
            if (YBF) {
                const uint2 u = *reinterpret_cast<const uint2*>((const bf16_t*)y_ + row + e);
                v[it] = make_float4(lo_f(u.x), hi_f(u.x), lo_f(u.y), hi_f(u.y));
            } else {
                v[it] = *reinterpret_cast<const float4*>((const float*)y_ + row + e);
            }
            s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
        }
    }
    if (resid) {
        const float mean = block_sum(s, red) / D;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int e = (tid + it * 256) * 4;
            if (e < D) {
                const float a = v[it].x - mean, b_ = v[it].y - mean, c = v[it].z - mean, d = v[it].w - mean;
                q += (a * a + b_ * b_) + (c * c + d * d);
            }
        }
        const float rstd = rsqrtf(block_sum(q, red) / D + eps);
        s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int e = (tid + it * 256) * 4;
            if (e < D) {
                const float4 wv = *reinterpret_cast<const float4*>(w + e), bv = *reinterpret_cast<const float4*>(b + e);
                const float4 rv = *reinterpret_cast<const float4*>(resid + row + e);
                v[it] = make_float4(rv.x + ((v[it].x - mean) * rstd * wv.x + bv.x), rv.y + ((v[it].y - mean) * rstd * wv.y + bv.y),
                                    rv.z + ((v[it].z - mean) * rstd * wv.z + bv.z), rv.w + ((v[it].w - mean) * rstd * wv.w + bv.w));
                *reinterpret_cast<float4*>(x_new + row + e) = v[it];
                s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
            }
        }
    }
    if (!w2) return;                                 // last block of the stack: no next pre-norm
    const float mean2 = block_sum(s, red) / D;
    float q2 = 0.f;
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int e = (tid + it * 256) * 4;
        if (e < D) {
            const float a = v[it].x - mean2, b_ = v[it].y - mean2, c = v[it].z - mean2, d = v[it].w - mean2;
            q2 += (a * a + b_ * b_) + (c * c + d * d);
        }
    }
    const float rstd2 = rsqrtf(block_sum(q2, red) / D + eps);
    int pos = 0, yq = 0, wq = 0;
    if (c_hi) {
        pos = pos_p[0];
        if (pos < 0 || pos >= cache_rows) return;
        if (pos > 0 && fmap > 0) { const int p = pos - 1; wq = p % fmap; yq = (p / fmap) % fmap; }
    }
    const size_t crow = ((size_t)bidx * cache_rows + pos) * D;
    const int qd = D >> 2;
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int e = (tid + it * 256) * 4;
        if (e >= D) continue;
        const float4 wv = *reinterpret_cast<const float4*>(w2 + e), bv = *reinterpret_cast<const float4*>(b2 + e);
        const float h[4] = {(v[it].x - mean2) * rstd2 * wv.x + bv.x, (v[it].y - mean2) * rstd2 * wv.y + bv.y,
                            (v[it].z - mean2) * rstd2 * wv.z + bv.z, (v[it].w - mean2) * rstd2 * wv.w + bv.w};
        bf16_t hh[4], hl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (o_lo) f2bf_hilo(h[i], hh[i], hl[i]);
            else { hh[i] = f2bf(h[i]); hl[i] = 0; }
        }
        uint2 oh = make_uint2(pack2(hh[0], hh[1]), pack2(hh[2], hh[3])), ol = make_uint2(pack2(hl[0], hl[1]), pack2(hl[2], hl[3]));
        if (c_hi) {
            *reinterpret_cast<uint2*>(c_hi + crow + e) = oh;
            if (c_lo) *reinterpret_cast<uint2*>(c_lo + crow + e) = ol;
            if (fmap > 0 && pos > 0 && e < 2 * qd) {  // D % 16 == 0: a 4-element group never straddles a quarter
                const bool up = e < qd, has = up ? (yq > 0) : (wq > 0);
                const size_t srow = crow - (size_t)(up ? fmap : 1) * D;
                oh = has ? *reinterpret_cast<const uint2*>(c_hi + srow + e) : make_uint2(0u, 0u);
                ol = (has && c_lo) ? *reinterpret_cast<const uint2*>(c_lo + srow + e) : make_uint2(0u, 0u);
            } else if (fmap < 0 && e < 2 * qd) {      // ShiftAudioTokens (np.py:157-183): the first half of the channels comes from
                const size_t srow = crow - (size_t)D; //  the previous row, <bos> included; zeros for row 0
                oh = pos > 0 ? *reinterpret_cast<const uint2*>(c_hi + srow + e) : make_uint2(0u, 0u);
                ol = (pos > 0 && c_lo) ? *reinterpret_cast<const uint2*>(c_lo + srow + e) : make_uint2(0u, 0u);
            }
        }
        *reinterpret_cast<uint2*>(o_hi + row + e) = oh;
        if (o_lo) *reinterpret_cast<uint2*>(o_lo + row + e) = ol;
    }
}

// ---- single-query attention pieces shared by the 3DNA and the text cross-attention decode kernels -------------------------
// q . k over DH (multiple of 8) bf16 values, 16 B per load; qrow: fp32 in LDS
template <int DHT>
__device__ __forceinline__ float dot_q_k(const float* qrow, const bf16_t* khi, const bf16_t* klo, int DH) {
    float acc = 0.f;
    if constexpr (DHT > 0) {                           // compile-time width: every 16 B load is in flight before the first FMA
        constexpr int NV = DHT > 0 ? DHT / 8 : 1;
        uint4 h[NV], l[NV];
#pragma unroll
        for (int v = 0; v < DHT / 8; ++v) h[v] = *reinterpret_cast<const uint4*>(khi + v * 8);
        if (klo) {
#pragma unroll
            for (int v = 0; v < DHT / 8; ++v) l[v] = *reinterpret_cast<const uint4*>(klo + v * 8);
        }
#pragma unroll
        for (int v = 0; v < DHT / 8; ++v) {
            float k[8] = {lo_f(h[v].x), hi_f(h[v].x), lo_f(h[v].y), hi_f(h[v].y), lo_f(h[v].z), hi_f(h[v].z), lo_f(h[v].w), hi_f(h[v].w)};
            if (klo) {
                k[0] += lo_f(l[v].x); k[1] += hi_f(l[v].x); k[2] += lo_f(l[v].y); k[3] += hi_f(l[v].y);
                k[4] += lo_f(l[v].z); k[5] += hi_f(l[v].z); k[6] += lo_f(l[v].w); k[7] += hi_f(l[v].w);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qrow[v * 8 + e], k[e], acc);
        }
        return acc;
    }
    for (int d = 0; d < DH; d += 8) {
        const uint4 h = *reinterpret_cast<const uint4*>(khi + d);
        float k[8] = {lo_f(h.x), hi_f(h.x), lo_f(h.y), hi_f(h.y), lo_f(h.z), hi_f(h.z), lo_f(h.w), hi_f(h.w)};
        if (klo) {
            const uint4 l = *reinterpret_cast<const uint4*>(klo + d);
            k[0] += lo_f(l.x); k[1] += hi_f(l.x); k[2] += lo_f(l.y); k[3] += hi_f(l.y);
            k[4] += lo_f(l.z); k[5] += hi_f(l.z); k[6] += lo_f(l.w); k[7] += hi_f(l.w);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(qrow[d + e], k[e], acc);
    }
    return acc;
}

// out[c0 .. c0+8) += sum over this thread's slots of P'[j] * V_j[c0 .. c0+8): 16 B loads, no branch (masked slots carry
// P' == 0 exactly and point at finite rows), eight loads in flight
template <typename RowOf>
__device__ __forceinline__ void pv_chunk(float (&acc)[8], const float* pmg, const bf16_t* vhi, const bf16_t* vlo, int j0, int jstep,
                                         int J, RowOf row_off) {
#pragma unroll 8
    for (int j = j0; j < J; j += jstep) {
        const float pj = pmg[j];
        const size_t off = row_off(j);
        const uint4 h = *reinterpret_cast<const uint4*>(vhi + off);
        float v[8] = {lo_f(h.x), hi_f(h.x), lo_f(h.y), hi_f(h.y), lo_f(h.z), hi_f(h.z), lo_f(h.w), hi_f(h.w)};
        if (vlo) {
            const uint4 l = *reinterpret_cast<const uint4*>(vlo + off);
            v[0] += lo_f(l.x); v[1] += hi_f(l.x); v[2] += lo_f(l.y); v[3] += hi_f(l.y);
            v[4] += lo_f(l.z); v[5] += hi_f(l.z); v[6] += lo_f(l.w); v[7] += hi_f(l.w);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, v[e], acc[e]);
    }
}

// the slot groups' partial sums meet in LDS: red [ngrp][inner], then one thread per channel writes the output
__device__ __forceinline__ void pv_finish(const float (&acc)[8], float* red, int grp, int ngrp, int c0, int inner, bf16_t* o,
                                          bf16_t* ol, size_t obase) {
    if (grp < ngrp)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[grp * inner + c0 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < inner; c += blockDim.x) {
        float t = 0.f;
        for (int gi = 0; gi < ngrp; ++gi) t += red[gi * inner + c];
        st_hl(o, ol, obase + c, t);
    }
}

// softmax over the J slots of every head (fp32, masked slots -> exactly 0; np.py:554 / 366), one wave per head round-robin;
// then the talking-heads mix pm[g][j] = sum_h W[g,h] P[h][j] (np.py:556-558 / 368-370).  s, pm: [NH][JS] in LDS
__device__ __forceinline__ void softmax_mix(float* s, float* pm, const int* ok, const float* wth, int NH, int J, int JS) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int h = wave; h < NH; h += nw) {
        float m = -3.4028234663852886e38f;
        for (int j = lane; j < J; j += 64) m = fmaxf(m, s[h * JS + j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < J; j += 64) {
            const float e = ok[j] ? __expf(s[h * JS + j] - m) : 0.f;
            s[h * JS + j] = e;
            sum += e;
        }
        const float inv = 1.f / wave_sum(sum);
        for (int j = lane; j < J; j += 64) s[h * JS + j] *= inv;
    }
    __syncthreads();
    for (int idx = tid; idx < NH * J; idx += blockDim.x) {
        const int g = idx / J, j = idx - g * J;
        float acc = 0.f;
        for (int h = 0; h < NH; ++h) acc = fmaf(wth[g * NH + h], s[h * JS + j], acc);
        pm[g * JS + j] = acc;
    }
    __syncthreads();
}

struct S3DecArgs {
    const bf16_t *qkv, *qkvl;        // new row [B, 3*inner]: q | k | v  (q unscaled)
    bf16_t *kv, *kvl;                // cache [B, cache_rows, 2*inner]: k | v
    bf16_t *o, *ol;                  // [B, inner]
    const float *wth, *rel;
    const int* pos;
    int cache_rows, H, W, kf, kh, kw, df, dh, dw, heads, dim_head, J;
    float scale;
};

// one workgroup (512 threads) per sample.
// LDS: q [NH][DH+1] | s [NH][J] | pm [NH][J] | krow [J] (key row per slot, -1 = masked) | ok [J] | red [4096]
template <int DHT>
__global__ __launch_bounds__(512) void s3_decode_kernel(S3DecArgs a) {
    extern __shared__ float sm[];
    const int inner = a.heads * a.dim_head, J = a.J, NH = a.heads, DH = a.dim_head, QS = DH + 1;
    float* qs = sm;
    float* s = qs + NH * QS;
    float* pm = s + NH * J;
    int* krow = reinterpret_cast<int*>(pm + NH * J);
    int* ok = krow + J;
    const int b = blockIdx.x, tid = threadIdx.x, pos = a.pos[0];
    if (pos < 0 || pos >= a.cache_rows) return;
    const size_t nrow = (size_t)b * 3 * inner;
    bf16_t* kvb = a.kv + (size_t)b * a.cache_rows * 2 * inner;
    bf16_t* kvlb = a.kvl ? a.kvl + (size_t)b * a.cache_rows * 2 * inner : nullptr;
    // 1. the new key / value row joins the cache
    for (int c = tid; c < 2 * inner; c += blockDim.x) {
        kvb[(size_t)pos * 2 * inner + c] = a.qkv[nrow + inner + c];
        if (kvlb) kvlb[(size_t)pos * 2 * inner + c] = a.qkvl[nrow + inner + c];
    }
    if (pos == 0) {                  // <bos> query: its output is its own value row (np.py:499, 608)
        for (int c = tid; c < inner; c += blockDim.x) {
            a.o[(size_t)b * inner + c] = a.qkv[nrow + 2 * inner + c];
            if (a.ol) a.ol[(size_t)b * inner + c] = a.qkvl ? a.qkvl[nrow + 2 * inner + c] : (bf16_t)0;
        }
        return;
    }
    for (int c = tid; c < inner; c += blockDim.x) qs[(c / DH) * QS + c % DH] = ld_hl(a.qkv, a.qkvl, nrow + c) * a.scale;
    // 2. key row of every slot: slot 0 is <bos>, slot 1 + (ta, tb, tc) the causal tap (np.py:420-457)
    const int p = pos - 1, w0 = p % a.W, y0 = (p / a.W) % a.H, f0 = p / (a.W * a.H);
    for (int j = tid; j < J; j += blockDim.x) {
        int r = 0;
        if (j > 0) {
            const int t = j - 1, tc = t % a.kw, tb = (t / a.kw) % a.kh, ta = t / (a.kw * a.kh);
            const int ff = f0 - (a.kf - 1 - ta) * a.df, yy = y0 - (a.kh - 1 - tb) * a.dh, ww = w0 - (a.kw - 1 - tc) * a.dw;
            r = (ff < 0 || yy < 0 || ww < 0) ? -1 : 1 + (ff * a.H + yy) * a.W + ww;
        }
        krow[j] = r;
        ok[j] = r >= 0;
    }
    __syncthreads();
    // 3. scores (fp32): consecutive threads take the heads of one key row (contiguous 2 * DH bytes each)
    for (int idx = tid; idx < NH * J; idx += blockDim.x) {
        const int j = idx / NH, h = idx - j * NH, r = krow[j];
        float sc = -3.4028234663852886e38f;
        if (r >= 0) {
            const size_t base = (size_t)r * 2 * inner + (size_t)h * DH;
            sc = dot_q_k<DHT>(qs + h * QS, kvb + base, kvlb ? kvlb + base : nullptr, DH) + ((a.rel && j > 0) ? a.rel[(size_t)j * NH + h] : 0.f);
        }
        s[h * J + j] = sc;
    }
    __syncthreads();
    softmax_mix(s, pm, ok, a.wth, NH, J, J);
    // 4. o[g] = sum_j P'[g, j] v_j[g]: thread = (8-channel chunk, slot group)
    {
        const int nchunk = inner / 8, ngrp = blockDim.x / nchunk;
        const int ch = tid % nchunk, grp = tid / nchunk, c0 = ch * 8, g = c0 / DH;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (grp < ngrp)
            pv_chunk(acc, pm + g * J, kvb, kvlb, grp, ngrp, J,
                     [&](int j) { return (size_t)max(krow[j], 0) * 2 * inner + inner + c0; });
        pv_finish(acc, reinterpret_cast<float*>(ok + J), grp, ngrp, c0, inner, a.o, a.ol, (size_t)b * inner);
    }
}

struct XDecArgs {
    const bf16_t *q, *ql;            // [B, ldq] unscaled
    const bf16_t *Kp, *Kpl, *Vp, *Vpl;   // [B][NH][JP][DH]
    const uint8_t* valid;            // [B][JP]
    bf16_t *o, *ol;                  // [B, ldo]
    const float* wth;
    int ldq, ldo, J, JP, heads, dim_head;
    float scale;
};

// text cross-attention of ONE query per sample (Attention.forward with context, np.py:339-378): null key at slot 0, key mask,
// fp32 softmax, talking heads; 1024 threads per sample (the work is a chain of memory latencies: width hides them).
// LDS: q [NH][DH+1] | s [NH][J] | pm [NH][J] | ok [J] | red [8192]
template <int DHT>
__global__ __launch_bounds__(1024) void xattn_decode_kernel(XDecArgs a) {
    extern __shared__ float sm[];
    const int NH = a.heads, DH = a.dim_head, J = a.J, JP = a.JP, inner = NH * DH, QS = DH + 1;
    float* qs = sm;
    float* s = qs + NH * QS;
    float* pm = s + NH * J;
    int* ok = reinterpret_cast<int*>(pm + NH * J);
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < inner; c += blockDim.x) qs[(c / DH) * QS + c % DH] = ld_hl(a.q, a.ql, (size_t)b * a.ldq + c) * a.scale;
    for (int j = tid; j < J; j += blockDim.x) ok[j] = a.valid[(size_t)b * JP + j] != 0;
    __syncthreads();
    const size_t kb = (size_t)b * NH * JP * DH;
    for (int idx = tid; idx < NH * J; idx += blockDim.x) {       // consecutive threads: consecutive key rows of one head
        const int h = idx / J, j = idx - h * J;
        float sc = -3.4028234663852886e38f;
        if (ok[j]) {
            const size_t base = kb + ((size_t)h * JP + j) * DH;
            sc = dot_q_k<DHT>(qs + h * QS, a.Kp + base, a.Kpl ? a.Kpl + base : nullptr, DH);
        }
        s[h * J + j] = sc;
    }
    __syncthreads();
    softmax_mix(s, pm, ok, a.wth, NH, J, J);
    // o[g][d] = sum_j P'[g][j] V[g][j][d]: thread = (8-channel chunk, slot group)
    {
        const int nchunk = inner / 8, ngrp = blockDim.x / nchunk;
        const int ch = tid % nchunk, grp = tid / nchunk, c0 = ch * 8, g = c0 / DH, d = c0 - g * DH;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const size_t vb = kb + (size_t)g * JP * DH + d;
        if (grp < ngrp) pv_chunk(acc, pm + g * J, a.Vp, a.Vpl, grp, ngrp, J, [&](int j) { return vb + (size_t)j * DH; });
        pv_finish(acc, reinterpret_cast<float*>(ok + J), grp, ngrp, c0, inner, a.o, a.ol, (size_t)b * a.ldo);
    }
}

}  // namespace

extern "C" int amdnuwa_decode_shift(const uint16_t* h_hi, const uint16_t* h_lo, uint16_t* cache_hi, uint16_t* cache_lo,
                                    uint16_t* out_hi, uint16_t* out_lo, const int* pos, int B, int cache_rows, int D,
                                    int fmap, hipStream_t stream) {
    if (!h_hi || !cache_hi || !out_hi || !pos || D <= 0 || D % 4 || fmap <= 0 || cache_rows <= 0) return AMDNUWA_ERR_ARG;
    if ((h_lo != nullptr) != (cache_lo != nullptr) || (out_lo && !h_lo)) return AMDNUWA_ERR_ARG;
    if (B <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(decode_shift_kernel, dim3(B), dim3(256), 0, stream, h_hi, h_lo, cache_hi, cache_lo, out_hi, out_lo, pos,
                       cache_rows, D, fmap);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_s3_decode(const amdnuwa_s3_geom* g, const uint16_t* qkv, const uint16_t* qkv_lo, uint16_t* kv_cache,
                                 uint16_t* kv_cache_lo, int cache_rows, const int* pos, const float* w_th, uint16_t* o,
                                 uint16_t* o_lo, hipStream_t stream) {
    if (!g || !qkv || !kv_cache || !pos || !w_th || !o) return AMDNUWA_ERR_ARG;
    if (g->heads <= 0 || g->dim_head <= 0 || g->kf <= 0 || g->kh <= 0 || g->kw <= 0 || g->df <= 0 || g->dh <= 0 || g->dw <= 0 ||
        g->F <= 0 || g->H <= 0 || g->W <= 0)
        return AMDNUWA_ERR_ARG;
    if ((qkv_lo != nullptr) != (kv_cache_lo != nullptr)) return AMDNUWA_ERR_ARG;
    if (g->noncausal) return AMDNUWA_ERR_UNSUPPORTED;          // incremental decoding needs every tap behind the query
    if (cache_rows < 1 || cache_rows > 1 + g->F * g->H * g->W) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    S3DecArgs a{};
    a.qkv = qkv; a.qkvl = qkv_lo; a.kv = kv_cache; a.kvl = kv_cache_lo; a.o = o; a.ol = o_lo; a.wth = w_th; a.rel = g->rel_bias;
    a.pos = pos; a.cache_rows = cache_rows; a.H = g->H; a.W = g->W; a.kf = g->kf; a.kh = g->kh; a.kw = g->kw;
    a.df = g->df; a.dh = g->dh; a.dw = g->dw; a.heads = g->heads; a.dim_head = g->dim_head; a.scale = g->scale;
    a.J = g->kf * g->kh * g->kw + 1;
    if (g->dim_head % 8 || g->heads * g->dim_head > 4096) return AMDNUWA_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)g->heads * (g->dim_head + 1) + 2 * (size_t)g->heads * a.J + 4096) * sizeof(float) + 2 * (size_t)a.J * sizeof(int);
    if (lds > 160 * 1024) return AMDNUWA_ERR_UNSUPPORTED;
#define S3D(DHT_) do { (void)hipFuncSetAttribute((const void*)s3_decode_kernel<DHT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                       hipLaunchKernelGGL((s3_decode_kernel<DHT_>), dim3(g->B), dim3(512), lds, stream, a); } while (0)
    if (g->dim_head == 64) S3D(64); else if (g->dim_head == 32) S3D(32); else S3D(0);
#undef S3D
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_xattn_decode(const amdnuwa_xattn_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq,
                                    const amdnuwa_xattn_kv* packed, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo,
                                    hipStream_t stream) {
    if (!g || !q || !packed || !packed->Kp || !packed->Vp || !packed->valid || !w_th || !o) return AMDNUWA_ERR_ARG;
    if (g->heads <= 0 || g->dim_head <= 0 || g->dim_head % 8 || g->T < 0 || g->JP < g->T + 1 || g->n != 1) return AMDNUWA_ERR_ARG;
    if ((q_lo != nullptr) != (packed->Kp_lo != nullptr) || (q_lo != nullptr) != (packed->Vp_lo != nullptr)) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    XDecArgs a{};
    a.q = q; a.ql = q_lo; a.Kp = packed->Kp; a.Kpl = packed->Kp_lo; a.Vp = packed->Vp; a.Vpl = packed->Vp_lo;
    a.valid = packed->valid; a.o = o; a.ol = o_lo; a.wth = w_th; a.ldq = ldq; a.ldo = ldo; a.J = g->T + 1; a.JP = g->JP;
    a.heads = g->heads; a.dim_head = g->dim_head; a.scale = g->scale;
    if (g->heads * g->dim_head > 8192) return AMDNUWA_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)g->heads * (g->dim_head + 1) + 2 * (size_t)g->heads * a.J + 8192) * sizeof(float) + (size_t)a.J * sizeof(int);
    if (lds > 160 * 1024) return AMDNUWA_ERR_UNSUPPORTED;
#define XD(DHT_) do { (void)hipFuncSetAttribute((const void*)xattn_decode_kernel<DHT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                      hipLaunchKernelGGL((xattn_decode_kernel<DHT_>), dim3(g->B), dim3(1024), lds, stream, a); } while (0)
    if (g->dim_head == 64) XD(64); else if (g->dim_head == 32) XD(32); else XD(0);
#undef XD
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_decode_ln(const void* y, int y_is_bf16, const float* resid, const float* w, const float* b,
                                 const float* next_w, const float* next_b, float* x_new, uint16_t* cache_hi, uint16_t* cache_lo,
                                 uint16_t* out_hi, uint16_t* out_lo, const int* pos, int B, int cache_rows, int D, int fmap,
                                 float eps, hipStream_t stream) {
    if (!y || D <= 0 || D % 4 || D > 4096) return AMDNUWA_ERR_ARG;
    if (resid && (!w || !b || !x_new)) return AMDNUWA_ERR_ARG;
    if (!resid && y_is_bf16) return AMDNUWA_ERR_ARG;                 // without a post-norm, y IS the fp32 stream row
    if (next_w && (!next_b || !out_hi)) return AMDNUWA_ERR_ARG;
    if (!next_w && !resid) return AMDNUWA_ERR_ARG;
    if (cache_hi && (!pos || fmap == 0 || fmap < -1 || cache_rows <= 0 || D % 16 || !next_w)) return AMDNUWA_ERR_ARG;
    if (cache_hi && ((cache_lo != nullptr) != (out_lo != nullptr))) return AMDNUWA_ERR_ARG;
    if (B <= 0) return AMDNUWA_OK;
    if (y_is_bf16)
        hipLaunchKernelGGL((decode_ln_kernel<true>), dim3(B), dim3(256), 0, stream, y, resid, w, b, next_w, next_b, x_new, cache_hi,
                           cache_lo, out_hi, out_lo, pos, cache_rows, D, fmap, eps);
    else
        hipLaunchKernelGGL((decode_ln_kernel<false>), dim3(B), dim3(256), 0, stream, y, resid, w, b, next_w, next_b, x_new, cache_hi,
                           cache_lo, out_hi, out_lo, pos, cache_rows, D, fmap, eps);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
