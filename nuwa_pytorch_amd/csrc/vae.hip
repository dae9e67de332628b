// Frozen VQGanVAE tokenizer path (VQGanVAE.get_video_indices -> encode, vq.py:431-435, 452-458) in exact fp32.
//
// VQ code indices must be bit-exact against the reference, so nothing here drops below fp32: the conv stack
// and the cosine-similarity search run on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bitwise an fp32 fmaf
// chain, 157 TFLOP/s peak = the fp32 vector rate, but it leaves the VALU free for the im2col addressing).
//
//   conv2d_fwd  : implicit-GEMM convolution  Y[n][co][oy][ox] = b[co] + sum_{ci,ky,kx} W[co][ci][ky][kx] X[n][ci][iy][ix]
//                 (encoders: 5x5 pad 2; 4x4 stride 2 pad 1 + LeakyReLU(0.1); ResBlock 3x3 / 1x1 -- vq.py:352-365, 228-242)
//                 GEMM view: M = Cout, N = batch*Ho*Wo pixels, K = Cin*KH*KW; 128x128 (or 64x256) tile, K-step 16, 4 waves of 64x64.
//   groupnorm   : nn.GroupNorm(16, C) (+ LeakyReLU) of ResBlock (vq.py:233-237)
//   vq_argmax   : idx = argmax_c <l2norm(x), l2norm(codebook[c])>  with the LOWEST index on exact ties
//                 (restated eval path of vector_quantize_pytorch -- PARITY UNPINNED)
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ConvArgs {
    const float *x, *w, *bias;
    float* y;
    int N, Cin, H, W, Cout, KH, KW, stride, pad, Ho, Wo, leaky;
    float slope;
};

// C/D layout of v_mfma_f32_32x32x2_f32: acc[reg] -> row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31
// A operand: lane holds A[row = lane & 31][k = lane >> 5];  B operand: lane holds B[k = lane >> 5][col = lane & 31]
constexpr int CK = 16;

// Tile TM output channels x TN pixels, 4 waves of 64 x 64 (2 x 2 MFMA tiles each): 128 x 128 for wide layers, 64 x 256 where
// Cout <= 64 (half of a 128-channel tile would multiply zeros).  KS = compile-time kernel size (0: read it from the arguments).
// The loader is the part that decides the speed of this kernel, not the MFMAs:
//   * every thread owns ONE pixel column and wave-uniform k rows, so the (ci, ky, kx) split of k is scalar work and, with KS known,
//     division by constants;
//   * weights [Cout][K] are read along K (float4 per thread, four lanes per 64-byte row piece): a wave touches 16 lines per load
//     where a per-channel column read touches 64;
//   * the next k-tile is fetched into registers under the current tile's MFMAs and stored to the other LDS buffer: one barrier per tile.
template <int KS, int TM, int TN>
__global__ __launch_bounds__(256) void conv2d_kernel(ConvArgs a) {
    constexpr int LDA = TM + 4, LDB = TN + 4;             // +4 pad: conflict-free operand reads
    constexpr int NA4 = TM / 64;                          // float4 weight pieces per thread and k-tile
    constexpr int NB = CK * TN / 256;                     // im2col elements per thread and k-tile
    __shared__ float As[2][CK][LDA];                      // weights  [k][co]
    __shared__ float Bs[2][CK][LDB];                      // im2col   [k][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = TM == 128 ? wave >> 1 : 0, wn = TM == 128 ? wave & 1 : wave;
    const int KH = KS ? KS : a.KH, KW = KS ? KS : a.KW, KHW = KH * KW;
    const int K = a.Cin * KHW;
    const int HoWo = a.Ho * a.Wo;
    const long long NP = (long long)a.N * HoWo;
    const int co0 = blockIdx.y * TM;
    const long long p0 = (long long)blockIdx.x * TN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // im2col side: pixel column lc, k rows kb + (256 / TN) * i
    const int lc = tid % TN;
    const int kb = __builtin_amdgcn_readfirstlane(tid / TN);
    const long long pix = p0 + lc;
    const bool pok = pix < NP;
    int pn = 0, oy = 0, ox = 0;
    if (pok) { pn = (int)(pix / HoWo); const int rem = (int)(pix % HoWo); oy = rem / a.Wo; ox = rem % a.Wo; }
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    const float* xb = a.x + (long long)pn * a.Cin * a.H * a.W + (long long)iy0 * a.W + ix0;      // only dereferenced where valid
    const int HW = a.H * a.W;
    // weight side: channel (tid >> 2) + 64 q, k piece 4 * (tid & 3)
    const int wco = tid >> 2, wk = 4 * (tid & 3);
    const bool w_vec = (K & 3) == 0;
    float4 wreg[NA4];
    float xreg[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NA4; ++q) {
            const int co = co0 + wco + 64 * q, k = k0 + wk;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < a.Cout) {
                const float* wp = a.w + (size_t)co * K + k;
                if (w_vec) { if (k < K) v = *reinterpret_cast<const float4*>(wp); }
                else {
                    if (k + 0 < K) v.x = wp[0];
                    if (k + 1 < K) v.y = wp[1];
                    if (k + 2 < K) v.z = wp[2];
                    if (k + 3 < K) v.w = wp[3];
                }
            }
            wreg[q] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = k0 + kb + (256 / TN) * i;           // wave-uniform
            const int ci = k / KHW, r = k - ci * KHW, ky = r / KW, kx = r - ky * KW;
            const bool ok = k < K && pok && (unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W;
            xreg[i] = ok ? xb[ci * HW + ky * a.W + kx] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NA4; ++q) {
            float* d = &As[buf][wk][wco + 64 * q];
            d[0] = wreg[q].x; d[LDA] = wreg[q].y; d[2 * LDA] = wreg[q].z; d[3 * LDA] = wreg[q].w;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][kb + (256 / TN) * i][lc] = xreg[i];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += CK, buf ^= 1) {
        const bool more = k0 + CK < K;
        if (more) fetch(k0 + CK);
#pragma unroll
        for (int kk = 0; kk < CK; kk += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[buf][kk + (lane >> 5)][wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[buf][kk + (lane >> 5)][wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long px = p0 + wn * 64 + j * 32 + (lane & 31);
            if (px >= NP) continue;
            const int n = (int)(px / HoWo), rem = (int)(px % HoWo);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (c >= a.Cout) continue;
                float v = acc[i][j][r] + (a.bias ? a.bias[c] : 0.f);
                if (a.leaky) v = v > 0.f ? v : v * a.slope;
                a.y[((size_t)n * a.Cout + c) * HoWo + rem] = v;
            }
        }
}

// GroupNorm over (C/G, H, W) per (n, group), optional LeakyReLU; one block per (n, group)
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int C, int HW, int G,
                                                        float eps, int leaky, float slope) {
    __shared__ float red[2][4];
    const int n = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G;
    const size_t base = ((size_t)n * C + (size_t)g * cpg) * HW;
    const int cnt = cpg * HW;
    float s = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) s += x[base + e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / cnt;
    float q = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) { const float d = x[base + e] - mean; q += d * d; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / cnt + eps);
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int c = g * cpg + e / HW;
        float v = (x[base + e] - mean) * rstd * w[c] + b[c];
        if (leaky) v = v > 0.f ? v : v * slope;
        y[base + e] = v;
    }
}

// rows of x [R][Dc] and codebook [Cn][Dc] are l2-normalised (F.normalize: v / max(||v||, 1e-12)), then
// idx[r] = argmax_c <xn[r], cn[c]>.  One workgroup = 64 rows; codes are scanned in increasing order in tiles of
// 64 with a strict '>' so the LOWEST index wins exact ties.  sims via the f32 MFMA (64x64 tile per wave pair).
__global__ __launch_bounds__(256) void vq_argmax_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                        long long* __restrict__ idx, float* __restrict__ best_sim,
                                                        long long R, int Cn, int Dc) {
    extern __shared__ float sm[];
    float* Xs = sm;                          // [64][Dc + 1] normalised rows
    float* Cs = Xs + 64 * (Dc + 1);          // [64][Dc + 1] normalised code tile
    float* bv = Cs + 64 * (Dc + 1);          // [4][64] per-wave best value
    int* bi = reinterpret_cast<int*>(bv + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long r0 = (long long)blockIdx.x * 64;
    // load + normalise the 64 rows (wave w handles rows w, w+4, ...)
    for (int rr = wave; rr < 64; rr += 4) {
        const long long r = r0 + rr;
        float ss = 0.f;
        for (int d = lane; d < Dc; d += 64) { const float v = r < R ? x[r * Dc + d] : 0.f; Xs[rr * (Dc + 1) + d] = v; ss += v * v; }
        ss = wave_sum(ss);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        for (int d = lane; d < Dc; d += 64) Xs[rr * (Dc + 1) + d] *= inv;
    }
    // each wave owns a 32x32 block of the 64x64 sim tile: rows 32*(wave>>1), cols 32*(wave&1)
    const int rb = (wave >> 1) * 32, cbk = (wave & 1) * 32;
    float best[16];
    int besti[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = -3.0e38f; besti[r] = 0; }
    for (int c0 = 0; c0 < Cn; c0 += 64) {
        __syncthreads();
        for (int cc = wave; cc < 64; cc += 4) {
            const int cidx = c0 + cc;
            float ss = 0.f;
            for (int d = lane; d < Dc; d += 64) { const float v = cidx < Cn ? cb[(size_t)cidx * Dc + d] : 0.f; Cs[cc * (Dc + 1) + d] = v; ss += v * v; }
            ss = wave_sum(ss);
            const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
            for (int d = lane; d < Dc; d += 64) Cs[cc * (Dc + 1) + d] *= inv;
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = 0; k < Dc; k += 2) {
            const float af = Xs[(rb + (lane & 31)) * (Dc + 1) + k + (lane >> 5)];
            const float bf = Cs[(cbk + (lane & 31)) * (Dc + 1) + k + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        // acc[reg]: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) of the block, col = lane & 31 -> code c0 + cbk + col
        const int code = c0 + cbk + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (code < Cn && acc[r] > best[r]) { best[r] = acc[r]; besti[r] = code; }
    }
    // reduce over the 32 lanes sharing a row (different codes): max value, lowest index on ties
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = best[r];
        int ix = besti[r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(ix, o, 64);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if ((lane & 31) == 0) {
            const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            bv[(wave & 1) * 64 + row] = v;
            bi[(wave & 1) * 64 + row] = ix;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const long long r = r0 + tid;
        if (r < R) {
            float v0 = bv[tid], v1 = bv[64 + tid];
            int i0 = bi[tid], i1 = bi[64 + tid];
            const bool take1 = v1 > v0 || (v1 == v0 && i1 < i0);
            idx[r] = take1 ? i1 : i0;
            if (best_sim) best_sim[r] = take1 ? v1 : v0;
        }
    }
}

// ---- VQ lookup, second form (code_dim 256): the 128 rows of a workgroup live in REGISTERS as pre-normalised MFMA A operands (a wave
// owns 32 rows: 128 registers), the codes stream through a double-buffered k-major LDS tile of 32 codes (scaled by their inverse norm
// on the way in: the same values F.normalize would store), and the code axis is cut in slices so that rows x slices fills the chip in
// whole rounds.  Per code tile a wave issues 128 back-to-back v_mfma_f32_32x32x2_f32 on one accumulator (issue interval = dependent
// latency = 64 cycles).  Ties: strict '>' inside a slice (codes scanned upwards) and across slices (combined upwards) -> lowest index.
__global__ __launch_bounds__(256) void vq_inv_norm_kernel(const float* __restrict__ cb, float* __restrict__ inv, int Cn, int Dc) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= Cn) return;
    float ss = 0.f;
    for (int d = lane; d < Dc; d += 64) { const float v = cb[(size_t)c * Dc + d]; ss += v * v; }
    ss = wave_sum(ss);
    if (lane == 0) inv[c] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
}

constexpr int VQ_TC = 32, VQ_LD = 33;        // codes per tile, LDS row stride (k-major [k][code], +1 pad)
template <int DC>
__global__ __launch_bounds__(256, 2) void vq_argmax2_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                            const float* __restrict__ inv_cn, float* __restrict__ part_v,
                                                            int* __restrict__ part_i, long long R, int Cn, int per_slice) {
    extern __shared__ float sm[];                     // [2][DC][VQ_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
    const long long r0 = (long long)blockIdx.x * 128 + wave * 32;
    const int cbeg = blockIdx.y * per_slice, cend = min(Cn, cbeg + per_slice);
    // A operands: a[kk] = xn[row (lane & 31)][2 kk + hi]
    float a[DC / 2];
    {
        const long long row = r0 + (lane & 31);
        const float* xr = x + (row < R ? row : R - 1) * DC + hi;
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < DC / 2; ++kk) { a[kk] = xr[2 * kk]; ss += a[kk] * a[kk]; }
        ss += __shfl_xor(ss, 32, 64);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int kk = 0; kk < DC / 2; ++kk) a[kk] *= inv;
    }
    // staging map: thread -> code (tid >> 3) of the tile, 8 pieces of 4 consecutive k at k = (tid & 7) * 4 + 32 j
    const int scode = tid >> 3, spart = (tid & 7) * 4;
    float4 pre[DC / 32];
    auto fetch = [&](int c0) {
        const int c = c0 + scode;
        const float sc = c < cend ? inv_cn[c] : 0.f;
        const float* src = cb + (size_t)(c < cend ? c : cbeg) * DC + spart;
#pragma unroll
        for (int j = 0; j < DC / 32; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(src + 32 * j);
            pre[j] = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        }
    };
    auto stash = [&](int buf) {
        float* base = sm + (size_t)buf * DC * VQ_LD + scode;
#pragma unroll
        for (int j = 0; j < DC / 32; ++j) {
            const int k = spart + 32 * j;
            base[(k + 0) * VQ_LD] = pre[j].x; base[(k + 1) * VQ_LD] = pre[j].y; base[(k + 2) * VQ_LD] = pre[j].z; base[(k + 3) * VQ_LD] = pre[j].w;
        }
    };
    float best[16];
    int besti[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = -3.0e38f; besti[r] = cbeg; }
    fetch(cbeg);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int c0 = cbeg; c0 < cend; c0 += VQ_TC, buf ^= 1) {
        const bool more = c0 + VQ_TC < cend;
        if (more) fetch(c0 + VQ_TC);                  // in flight under the MFMAs below
        const float* bt = sm + (size_t)buf * DC * VQ_LD + hi * VQ_LD + (lane & 31);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DC / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bt[2 * kk * VQ_LD], acc, 0, 0, 0);
        const int code = c0 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (code < cend && acc[r] > best[r]) { best[r] = acc[r]; besti[r] = code; }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }
    // lanes sharing a row hold different codes: max value, lowest index on ties
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = best[r];
        int ix = besti[r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(ix, o, 64);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        const long long row = r0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if ((lane & 31) == 0 && row < R) { part_v[(size_t)blockIdx.y * R + row] = v; part_i[(size_t)blockIdx.y * R + row] = ix; }
    }
}
__global__ __launch_bounds__(256) void vq_combine_kernel(const float* __restrict__ part_v, const int* __restrict__ part_i, int S,
                                                         long long R, long long* __restrict__ idx, float* __restrict__ best_sim) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float v = part_v[r];
    int ix = part_i[r];
    for (int s = 1; s < S; ++s) {                     // slices hold increasing code ranges: strict '>' keeps the lowest index
        const float ov = part_v[(size_t)s * R + r];
        if (ov > v) { v = ov; ix = part_i[(size_t)s * R + r]; }
    }
    idx[r] = ix;
    if (best_sim) best_sim[r] = v;
}
static int vq_slices(long long R, int Cn) {
    const long long rb = (R + 127) / 128;
    int s = (int)((1024 + rb - 1) / rb);              // aim at >= 1024 workgroups (two resident per CU)
    const int smax = (Cn + 4 * VQ_TC - 1) / (4 * VQ_TC);      // at least 4 code tiles per slice
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

// ---- VQGanAttention core (vq.py:244-286), exact fp32 ----------------------------------------------------------------
// rows of length len: x <- x / max(||x||_2, 1e-12)   (F.normalize over the SPATIAL axis of q and k, quirk Q9)
// (rows come in `groups` groups of rows_per_group consecutive rows, group g starting at row g * group_stride_rows)
__global__ __launch_bounds__(256) void rows_l2norm_kernel(float* __restrict__ x, long long rows, int rows_per_group, int group_stride_rows, int len) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* p = x + ((r / rows_per_group) * group_stride_rows + r % rows_per_group) * len;
    float ss = 0.f;
    for (int i = lane; i < len; i += 64) ss += p[i] * p[i];
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int i = lane; i < len; i += 64) p[i] *= inv;
}

// one workgroup per (image, head), one thread per query position i (P <= 256 positions, c <= 64 channels per head):
//   s_ij = (sum_c qn[c][i] kn[c][j]) * exp(scale[head]) + bias[head][i][j];  softmax over j;  out[c][i] = sum_j p_ij v[c][j]
// kn and v of the head sit in LDS and are read by all threads at the same address (broadcast).  Two passes over j (row max,
// then exp / sum / PV) -- the whole block is < 0.5 % of the tokenizer, clarity over speed.
__global__ __launch_bounds__(256) void vqattn_core_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, float* __restrict__ out, int heads, int c,
                                                          int P) {
    extern __shared__ float sm[];
    float* ks = sm;                     // [c][P]
    float* vs = sm + c * P;             // [c][P]
    const int n = blockIdx.x / heads, hh = blockIdx.x % heads, i = threadIdx.x;
    const size_t img = (size_t)n * 3 * heads * c * P;
    const float* q = qkv + img + (size_t)hh * c * P;
    const float* k = qkv + img + (size_t)(heads + hh) * c * P;
    const float* v = qkv + img + (size_t)(2 * heads + hh) * c * P;
    for (int e = threadIdx.x; e < c * P; e += blockDim.x) { ks[e] = k[e]; vs[e] = v[e]; }
    __syncthreads();
    if (i >= P) return;
    float qv[64], acc[64];
#pragma unroll
    for (int cc = 0; cc < 64; ++cc) { qv[cc] = cc < c ? q[(size_t)cc * P + i] : 0.f; acc[cc] = 0.f; }
    const float se = expf(scale[hh]);
    const float* brow = bias + ((size_t)hh * P + i) * P;
    float m = -3.0e38f, l = 0.f;
    if (P % 4 == 0) {
        // 4 keys per step: one broadcast float4 LDS read feeds 4 FMAs
        for (int j = 0; j < P; j += 4) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int cc = 0; cc < 64; ++cc)
                if (cc < c) {
                    const float4 k4 = *reinterpret_cast<const float4*>(ks + cc * P + j);
                    d0 = fmaf(qv[cc], k4.x, d0); d1 = fmaf(qv[cc], k4.y, d1); d2 = fmaf(qv[cc], k4.z, d2); d3 = fmaf(qv[cc], k4.w, d3);
                }
            const float4 b4 = *reinterpret_cast<const float4*>(brow + j);
            m = fmaxf(fmaxf(m, fmaxf(d0 * se + b4.x, d1 * se + b4.y)), fmaxf(d2 * se + b4.z, d3 * se + b4.w));
        }
        for (int j = 0; j < P; j += 4) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int cc = 0; cc < 64; ++cc)
                if (cc < c) {
                    const float4 k4 = *reinterpret_cast<const float4*>(ks + cc * P + j);
                    d0 = fmaf(qv[cc], k4.x, d0); d1 = fmaf(qv[cc], k4.y, d1); d2 = fmaf(qv[cc], k4.z, d2); d3 = fmaf(qv[cc], k4.w, d3);
                }
            const float4 b4 = *reinterpret_cast<const float4*>(brow + j);
            const float p0 = expf(d0 * se + b4.x - m), p1 = expf(d1 * se + b4.y - m), p2 = expf(d2 * se + b4.z - m), p3 = expf(d3 * se + b4.w - m);
            l += (p0 + p1) + (p2 + p3);
#pragma unroll
            for (int cc = 0; cc < 64; ++cc)
                if (cc < c) {
                    const float4 v4 = *reinterpret_cast<const float4*>(vs + cc * P + j);
                    acc[cc] = fmaf(p3, v4.w, fmaf(p2, v4.z, fmaf(p1, v4.y, fmaf(p0, v4.x, acc[cc]))));
                }
        }
    } else {
        for (int j = 0; j < P; ++j) {
            float d = 0.f;
#pragma unroll
            for (int cc = 0; cc < 64; ++cc) if (cc < c) d = fmaf(qv[cc], ks[cc * P + j], d);
            m = fmaxf(m, d * se + brow[j]);
        }
        for (int j = 0; j < P; ++j) {
            float d = 0.f;
#pragma unroll
            for (int cc = 0; cc < 64; ++cc) if (cc < c) d = fmaf(qv[cc], ks[cc * P + j], d);
            const float pj = expf(d * se + brow[j] - m);
            l += pj;
#pragma unroll
            for (int cc = 0; cc < 64; ++cc) if (cc < c) acc[cc] = fmaf(pj, vs[cc * P + j], acc[cc]);
        }
    }
    const float il = 1.f / l;
    float* o = out + ((size_t)n * heads + hh) * c * P;
#pragma unroll
    for (int cc = 0; cc < 64; ++cc) if (cc < c) o[(size_t)cc * P + i] = acc[cc] * il;
}

// The same block on the f32 MFMA for the cfg-3 shape (64 channels per head, 256 positions).  Workgroup = (image, head, half of the
// queries), wave = 32 queries.  Everything is computed TRANSPOSED so that no operand ever needs a shuffle or an LDS round trip:
//   S^T[j][i] = sum_c k[c][j] q[c][i]   A = k from LDS ([c][j]: lanes along j), B = q held in 32 registers ([c][i]: lanes along i)
//   a lane then owns ONE query (i = lane & 31) and 128 of its 256 scores: the softmax reductions are in-lane plus one xor-32 exchange
//   O^T[c][i] = sum_j v[c][j] P^T[j][i]   B = the score registers as they are (a k-step pairs key j of the low half-wave with key
//   j + 4 of the high one; A reads v at the same pairing), A = v from LDS with row stride 257 (lanes along c: conflict-free)
constexpr int VA_C = 64, VA_P = 256, VA_LDV = 257;
__global__ __launch_bounds__(256) void vqattn_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, float* __restrict__ out, int heads) {
    extern __shared__ float sm[];
    float* ks = sm;                         // [64][256]
    float* vs = sm + VA_C * VA_P;           // [64][257]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, ln = lane & 31;
    const int nh = blockIdx.x >> 1, n = nh / heads, hh = nh % heads;
    const size_t img = (size_t)n * 3 * heads * VA_C * VA_P;
    const float* q = qkv + img + (size_t)hh * VA_C * VA_P;
    const float* k = qkv + img + (size_t)(heads + hh) * VA_C * VA_P;
    const float* v = qkv + img + (size_t)(2 * heads + hh) * VA_C * VA_P;
    for (int e = tid; e < VA_C * VA_P / 4; e += 256) {
        const int cc = e >> 6, j4 = (e & 63) * 4;
        *reinterpret_cast<float4*>(ks + cc * VA_P + j4) = *reinterpret_cast<const float4*>(k + cc * VA_P + j4);
        const float4 v4 = *reinterpret_cast<const float4*>(v + cc * VA_P + j4);
        float* d = vs + cc * VA_LDV + j4;
        d[0] = v4.x; d[1] = v4.y; d[2] = v4.z; d[3] = v4.w;
    }
    const int i = (blockIdx.x & 1) * 128 + wave * 32 + ln;
    float qb[VA_C / 2];
#pragma unroll
    for (int kk = 0; kk < VA_C / 2; ++kk) qb[kk] = q[(2 * kk + hi) * VA_P + i];
    __syncthreads();
    f32x16 st[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[T][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < VA_C / 2; ++kk)
            st[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(2 * kk + hi) * VA_P + 32 * T + ln], qb[kk], st[T], 0, 0, 0);
    }
    // st[T][r] = s(i, j) with j = 32 T + (r & 3) + 8 (r >> 2) + 4 hi: four consecutive keys per register quad
    const float se = expf(scale[hh]);
    const float* brow = bias + ((size_t)hh * VA_P + i) * VA_P;
    float m = -3.0e38f;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(brow + 32 * T + 8 * g + 4 * hi);
            st[T][4 * g + 0] = st[T][4 * g + 0] * se + b4.x; st[T][4 * g + 1] = st[T][4 * g + 1] * se + b4.y;
            st[T][4 * g + 2] = st[T][4 * g + 2] * se + b4.z; st[T][4 * g + 3] = st[T][4 * g + 3] * se + b4.w;
            m = fmaxf(fmaxf(m, fmaxf(st[T][4 * g + 0], st[T][4 * g + 1])), fmaxf(st[T][4 * g + 2], st[T][4 * g + 3]));
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[T][r] = expf(st[T][r] - m); l += st[T][r]; }
    l += __shfl_xor(l, 32, 64);
    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * T + (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int t = 0; t < 2; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(32 * t + ln) * VA_LDV + j], st[T][r], o[t], 0, 0, 0);
        }
    const float il = 1.f / l;
    float* op = out + ((size_t)n * heads + hh) * VA_C * VA_P + i;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) op[(size_t)(32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi) * VA_P] = o[t][r] * il;
}

// LayerNormChan (vq.py:178-190) over the channel axis of an NCHW tensor, + residual.  A workgroup = 64 consecutive positions of one
// image x 4 channel quarters: each thread keeps its C / 4 values in registers (one HBM read), statistics meet in LDS.
template <int CQ>
__global__ __launch_bounds__(256) void chan_layernorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ b, const float* __restrict__ resid,
                                                                 float* __restrict__ y, int HW, float eps) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, C = 4 * CQ;
    const int per_img = HW / 64, n = blockIdx.x / per_img, p = (blockIdx.x % per_img) * 64 + lane;
    const size_t base = ((size_t)n * C + (size_t)part * CQ) * HW + p;
    float v[CQ];
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < CQ; ++ch) { v[ch] = x[base + (size_t)ch * HW]; s += v[ch]; }
    red[0][part][lane] = s;
    __syncthreads();
    const float mean = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) / C;
    float q = 0.f;
#pragma unroll
    for (int ch = 0; ch < CQ; ++ch) { const float d = v[ch] - mean; q += d * d; }
    red[1][part][lane] = q;
    __syncthreads();
    const float den = sqrtf(((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) / C + eps);
#pragma unroll
    for (int ch = 0; ch < CQ; ++ch) {
        const int c = part * CQ + ch;
        float r = (v[ch] - mean) / den * g[c] + b[c];
        if (resid) r += resid[base + (size_t)ch * HW];
        y[base + (size_t)ch * HW] = r;
    }
}

// LayerNormChan, any shape: one thread per (image, position)
__global__ __launch_bounds__(256) void chan_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                             const float* __restrict__ resid, float* __restrict__ y, long long NP, int C, int HW,
                                                             float eps) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NP) return;
    const size_t base = (size_t)(t / HW) * C * HW + (size_t)(t % HW);
    float s = 0.f;
    for (int ch = 0; ch < C; ++ch) s += x[base + (size_t)ch * HW];
    const float mean = s / C;
    float q = 0.f;
    for (int ch = 0; ch < C; ++ch) { const float d = x[base + (size_t)ch * HW] - mean; q += d * d; }
    const float den = sqrtf(q / C + eps);                 // (x - mean) / (var + eps).sqrt() * g + b, var unbiased=False
    for (int ch = 0; ch < C; ++ch) {
        float v = (x[base + (size_t)ch * HW] - mean) / den * g[ch] + b[ch];
        if (resid) v += resid[base + (size_t)ch * HW];
        y[base + (size_t)ch * HW] = v;
    }
}

// ---- decoder-only pieces (VQGanVAE.decode, vq.py:437-441; GLUResBlock vq.py:212-226) ---------------------------------
// nn.GLU(dim=1) on [N][2C][HW]: y[n][c] = x[n][c] * sigmoid(x[n][C + c])
__global__ __launch_bounds__(256) void glu_chan_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int C, int HW) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long n = t / ((long long)C * HW), r = t % ((long long)C * HW);
        const float a = x[n * 2 * C * HW + r], g = x[n * 2 * C * HW + (long long)C * HW + r];
        y[t] = a / (1.f + expf(-g));
    }
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) on NCHW: src = (dst + 0.5) / 2 - 0.5, clamped at 0
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = planes * Ho * Wo;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long pl = t / ((long long)Ho * Wo);
        const int oy = (int)((t / Wo) % Ho), ox = (int)(t % Wo);
        const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx, y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = sy - y0, lx = sx - x0;
        const float* p = x + pl * H * W;
        const float top = p[y0 * W + x0] * (1.f - lx) + p[y0 * W + x1] * lx, bot = p[y1 * W + x0] * (1.f - lx) + p[y1 * W + x1] * lx;
        y[t] = top * (1.f - ly) + bot * ly;
    }
}

}  // namespace

extern "C" int amdnuwa_glu_chan(const float* x, float* y, int N, int C, int HW, hipStream_t stream) {
    if (!x || !y || C <= 0 || HW <= 0) return AMDNUWA_ERR_ARG;
    const long long total = (long long)N * C * HW;
    if (total <= 0) return AMDNUWA_OK;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(glu_chan_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, stream, x, y, total, C, HW);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_upsample_bilinear2x(const float* x, float* y, int N, int C, int H, int W, hipStream_t stream) {
    if (!x || !y || H <= 0 || W <= 0) return AMDNUWA_ERR_ARG;
    const long long planes = (long long)N * C, total = planes * 4 * H * W;
    if (total <= 0) return AMDNUWA_OK;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, stream, x, y, planes, H, W);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_rows_l2norm(float* x, int groups, int rows_per_group, int group_stride_rows, int len, hipStream_t stream) {
    if (!x || len <= 0 || rows_per_group <= 0 || group_stride_rows < rows_per_group) return AMDNUWA_ERR_ARG;
    const long long rows = (long long)groups * rows_per_group;
    if (rows <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(rows_l2norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, rows, rows_per_group, group_stride_rows, len);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_vqattn_core(const float* qkv, const float* bias, const float* scale, float* out, int N, int heads, int dim_head,
                                   int P, hipStream_t stream) {
    if (!qkv || !bias || !scale || !out || heads <= 0) return AMDNUWA_ERR_ARG;
    if (dim_head < 1 || dim_head > 64 || P < 1 || P > 256) return AMDNUWA_ERR_UNSUPPORTED;
    if (N <= 0) return AMDNUWA_OK;
    if (dim_head == VA_C && P == VA_P && g_amdnuwa_tuning[15] != 1) {
        const size_t lds2 = (size_t)(VA_C * VA_P + VA_C * VA_LDV) * sizeof(float);
        (void)hipFuncSetAttribute((const void*)vqattn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(vqattn_mfma_kernel, dim3(2 * N * heads), dim3(256), lds2, stream, qkv, bias, scale, out, heads);
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    const size_t lds = (size_t)2 * dim_head * P * sizeof(float);
    (void)hipFuncSetAttribute((const void*)vqattn_core_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vqattn_core_kernel, dim3(N * heads), dim3(256), lds, stream, qkv, bias, scale, out, heads, dim_head, P);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_chan_layernorm(const float* x, const float* g, const float* b, const float* resid, float* y, int N, int C, int HW,
                                      float eps, hipStream_t stream) {
    if (!x || !g || !b || !y || C <= 0 || HW <= 0) return AMDNUWA_ERR_ARG;
    if (N <= 0) return AMDNUWA_OK;
    const long long NP = (long long)N * HW;
    if (HW % 64 == 0 && g_amdnuwa_tuning[15] != 1 && (C == 512 || C == 256 || C == 128 || C == 64)) {
        const dim3 grid((unsigned)(NP / 64));
        if (C == 512) hipLaunchKernelGGL((chan_layernorm_reg_kernel<128>), grid, dim3(256), 0, stream, x, g, b, resid, y, HW, eps);
        else if (C == 256) hipLaunchKernelGGL((chan_layernorm_reg_kernel<64>), grid, dim3(256), 0, stream, x, g, b, resid, y, HW, eps);
        else if (C == 128) hipLaunchKernelGGL((chan_layernorm_reg_kernel<32>), grid, dim3(256), 0, stream, x, g, b, resid, y, HW, eps);
        else hipLaunchKernelGGL((chan_layernorm_reg_kernel<16>), grid, dim3(256), 0, stream, x, g, b, resid, y, HW, eps);
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    hipLaunchKernelGGL(chan_layernorm_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, stream, x, g, b, resid, y, NP, C, HW, eps);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_conv2d_fwd(const amdnuwa_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                                  hipStream_t stream) {
    if (!d || !x || !w || !y) return AMDNUWA_ERR_ARG;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return AMDNUWA_ERR_ARG;
    const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    if (Ho != d->Ho || Wo != d->Wo || Ho <= 0 || Wo <= 0) return AMDNUWA_ERR_ARG;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad = d->pad; a.Ho = Ho; a.Wo = Wo; a.leaky = d->leaky; a.slope = 0.1f;
    const long long NP = (long long)d->N * Ho * Wo;
    const int ks = (d->KH == d->KW && (d->KH == 1 || d->KH == 3 || d->KH == 4 || d->KH == 5)) ? d->KH : 0;
    const bool narrow = d->Cout <= 64;
#define CONV_GO(KS_)                                                                                                                \
    do {                                                                                                                             \
        if (narrow)                                                                                                                  \
            hipLaunchKernelGGL((conv2d_kernel<KS_, 64, 256>), dim3((unsigned)((NP + 255) / 256), (d->Cout + 63) / 64), dim3(256), 0, \
                               stream, a);                                                                                           \
        else                                                                                                                         \
            hipLaunchKernelGGL((conv2d_kernel<KS_, 128, 128>), dim3((unsigned)((NP + 127) / 128), (d->Cout + 127) / 128), dim3(256), \
                               0, stream, a);                                                                                        \
    } while (0)
    switch (ks) {
        case 1: CONV_GO(1); break;
        case 3: CONV_GO(3); break;
        case 4: CONV_GO(4); break;
        case 5: CONV_GO(5); break;
        default: CONV_GO(0); break;
    }
#undef CONV_GO
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_groupnorm_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int HW, int groups,
                                     float eps, int leaky, hipStream_t stream) {
    if (!x || !w || !b || !y || groups <= 0 || C % groups) return AMDNUWA_ERR_ARG;
    if (N <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(groupnorm_kernel, dim3(N * groups), dim3(256), 0, stream, x, w, b, y, C, HW, groups, eps, leaky, 0.1f);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_vq_argmax(const float* x, const float* codebook, long long* indices, float* best_sim, long long R,
                                 int n_codes, int code_dim, hipStream_t stream) {
    if (!x || !codebook || !indices || n_codes <= 0 || code_dim <= 0 || code_dim % 2) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    const size_t lds = ((size_t)2 * 64 * (code_dim + 1) + 256) * sizeof(float) + 128 * sizeof(int);
    if (lds > 160 * 1024) return AMDNUWA_ERR_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)vq_argmax_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vq_argmax_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), lds, stream, x, codebook, indices, best_sim, R,
                       n_codes, code_dim);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_vq_argmax_workspace_bytes(long long R, int n_codes) {
    if (R <= 0 || n_codes <= 0) return 0;
    return ((size_t)n_codes + (size_t)vq_slices(R, n_codes) * (size_t)R * 2) * sizeof(float) + 64;
}

// as amdnuwa_vq_argmax, with a workspace: code_dim 256 runs the register-resident-rows kernel (~6x faster at the cfg-3 size),
// anything else the first form
extern "C" int amdnuwa_vq_argmax_ws(const float* x, const float* codebook, long long* indices, float* best_sim, long long R,
                                    int n_codes, int code_dim, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (code_dim != 256 || g_amdnuwa_tuning[15] == 1) return amdnuwa_vq_argmax(x, codebook, indices, best_sim, R, n_codes, code_dim, stream);
    if (!x || !codebook || !indices || n_codes <= 0) return AMDNUWA_ERR_ARG;
    if (R <= 0) return AMDNUWA_OK;
    if (!workspace || workspace_bytes < amdnuwa_vq_argmax_workspace_bytes(R, n_codes)) return AMDNUWA_ERR_WORKSPACE;
    const int S = vq_slices(R, n_codes);
    const int per_slice = ((n_codes + S - 1) / S + VQ_TC - 1) / VQ_TC * VQ_TC;
    float* inv = (float*)workspace;
    float* part_v = inv + n_codes;
    int* part_i = (int*)(part_v + (size_t)S * R);
    hipLaunchKernelGGL(vq_inv_norm_kernel, dim3((n_codes + 3) / 4), dim3(256), 0, stream, codebook, inv, n_codes, code_dim);
    LAUNCH_CHECK();
    const size_t lds = (size_t)2 * 256 * VQ_LD * sizeof(float);
    (void)hipFuncSetAttribute((const void*)vq_argmax2_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((vq_argmax2_kernel<256>), dim3((unsigned)((R + 127) / 128), S), dim3(256), lds, stream, x, codebook, inv, part_v,
                       part_i, R, n_codes, per_slice);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_combine_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, part_v, part_i, S, R, indices, best_sim);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
