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

struct S3DecArgs {
    const bf16_t *qkv, *qkvl;        // new row [B, 3*inner]: q | k | v  (q unscaled)
    bf16_t *kv, *kvl;                // cache [B, cache_rows, 2*inner]: k | v
    bf16_t *o, *ol;                  // [B, inner]
    const float *wth, *rel;
    const int* pos;
    int cache_rows, H, W, kf, kh, kw, df, dh, dw, heads, dim_head, J;
    float scale;
};

// one workgroup per sample.  LDS: q (inner floats) | s (heads*J) | pm (heads*J) | krow (J ints)
__global__ __launch_bounds__(256) void s3_decode_kernel(S3DecArgs a) {
    extern __shared__ float sm[];
    const int inner = a.heads * a.dim_head, J = a.J, NH = a.heads, DH = a.dim_head;
    float* qs = sm;
    float* s = qs + inner;
    float* pm = s + NH * J;
    int* krow = reinterpret_cast<int*>(pm + NH * J);
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
    for (int c = tid; c < inner; c += blockDim.x) qs[c] = ld_hl(a.qkv, a.qkvl, nrow + c) * a.scale;
    // 2. key row of every slot (-1 = masked): slot 0 is <bos>, slot 1 + (ta, tb, tc) the causal tap (np.py:420-457)
    const int p = pos - 1, w0 = p % a.W, y0 = (p / a.W) % a.H, f0 = p / (a.W * a.H);
    for (int j = tid; j < J; j += blockDim.x) {
        int r = 0;
        if (j > 0) {
            const int t = j - 1, tc = t % a.kw, tb = (t / a.kw) % a.kh, ta = t / (a.kw * a.kh);
            const int ff = f0 - (a.kf - 1 - ta) * a.df, yy = y0 - (a.kh - 1 - tb) * a.dh, ww = w0 - (a.kw - 1 - tc) * a.dw;
            r = (ff < 0 || yy < 0 || ww < 0) ? -1 : 1 + (ff * a.H + yy) * a.W + ww;
        }
        krow[j] = r;
    }
    __syncthreads();
    // 3. scores (fp32)
    for (int idx = tid; idx < NH * J; idx += blockDim.x) {
        const int j = idx / NH, h = idx % NH, r = krow[j];
        float sc = -3.4028234663852886e38f;
        if (r >= 0) {
            const size_t base = (size_t)r * 2 * inner + (size_t)h * DH;
            float acc = 0.f;
            for (int d = 0; d < DH; ++d) acc += qs[h * DH + d] * ld_hl(kvb, kvlb, base + d);
            sc = acc + ((a.rel && j > 0) ? a.rel[(size_t)j * NH + h] : 0.f);
        }
        s[h * J + j] = sc;
    }
    __syncthreads();
    // 4. softmax over the slots of each head (fp32, np.py:554)
    if (tid < NH) {
        float m = -3.4028234663852886e38f;
        for (int j = 0; j < J; ++j) m = fmaxf(m, s[tid * J + j]);
        float sum = 0.f;
        for (int j = 0; j < J; ++j) { const float e = (krow[j] >= 0) ? __expf(s[tid * J + j] - m) : 0.f; s[tid * J + j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < J; ++j) s[tid * J + j] *= inv;
    }
    __syncthreads();
    // 5. talking heads: P'[g] = sum_h W[g,h] P[h]  (np.py:556-558)
    for (int idx = tid; idx < NH * J; idx += blockDim.x) {
        const int g = idx / J, j = idx % J;
        float acc = 0.f;
        for (int h = 0; h < NH; ++h) acc += a.wth[g * NH + h] * s[h * J + j];
        pm[idx] = acc;
    }
    __syncthreads();
    // 6. o[g] = sum_j P'[g, j] v_j[g]
    for (int c = tid; c < inner; c += blockDim.x) {
        const int g = c / DH;
        float acc = 0.f;
        for (int j = 0; j < J; ++j) {
            const int r = krow[j];
            if (r >= 0) acc += pm[g * J + j] * ld_hl(kvb, kvlb, (size_t)r * 2 * inner + inner + c);
        }
        st_hl(a.o, a.ol, (size_t)b * inner + c, acc);
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
    if (cache_rows < 1 || cache_rows > 1 + g->F * g->H * g->W) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    S3DecArgs a{};
    a.qkv = qkv; a.qkvl = qkv_lo; a.kv = kv_cache; a.kvl = kv_cache_lo; a.o = o; a.ol = o_lo; a.wth = w_th; a.rel = g->rel_bias;
    a.pos = pos; a.cache_rows = cache_rows; a.H = g->H; a.W = g->W; a.kf = g->kf; a.kh = g->kh; a.kw = g->kw;
    a.df = g->df; a.dh = g->dh; a.dw = g->dw; a.heads = g->heads; a.dim_head = g->dim_head; a.scale = g->scale;
    a.J = g->kf * g->kh * g->kw + 1;
    const size_t lds = ((size_t)g->heads * g->dim_head + 2 * (size_t)g->heads * a.J) * sizeof(float) + (size_t)a.J * sizeof(int);
    if (lds > 160 * 1024) return AMDNUWA_ERR_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)s3_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(s3_decode_kernel, dim3(g->B), dim3(256), lds, stream, a);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
