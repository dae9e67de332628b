// Text cross-attention core of the decoder (Attention.forward with context, np.py:339-378) on MFMA.
//
//   keys j = 0..T: j = 0 is the learned null key/value (np.py:339-343), always visible (np.py:360);
//   sim = (q*scale) k^T, masked_fill(~context_mask, -fp32max), fp32 softmax, talking heads
//   Conv2d(h,h,1) ACROSS heads after the softmax (np.py:371-372), then attn @ v.
//
// Talking heads couples all heads of a (query, key) pair, so one workgroup = 32 queries x ALL heads
// (wave = head).  Each wave computes S^T = K Q^T with v_mfma_f32_16x16x32_bf16 in the "swapped"
// form: lane (c = lane&15, g = lane>>4) holds S^T[key = 16kb + 4g + r][query = 16qb + c], i.e. a
// query's score row lives in 4 lanes x registers -> row max / sum = register reduce + 2 shuffles.
// Normalised P (fp32) is exchanged between the waves through LDS in 32-key chunks, mixed with the
// h x h weight in fp32, and the mixed P' goes straight back into an MFMA as the B operand of
// O^T = V^T P'^T: the reduction index (key) may be permuted freely, so the C-layout registers
// (keys 4g+r and 16+4g+r) are used as k-slots (g, j) and V^T is fetched in the same permuted order.
//
// The backward saves P and P' (bf16 hi[/lo]) in the forward; bwd_q computes dP' = dO V^T, mixes it
// with W^T, forms ds = P (dP - sum_j P dP), dq = scale * ds K, and writes ds; dK / dV are batched
// TN GEMMs (reduction over the 2560 queries) on ds / P' -- no atomics, deterministic.
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

constexpr int MAXKB = 18;        // key blocks of 16: T + 1 <= 288
constexpr float NEG_MAX = -3.4028234663852886e38f;

struct XArgs {
    const bf16_t *q, *ql; int ldq;                 // [B*n, ldq], head h at cols h*DH
    const bf16_t *Kp, *Kpl, *Kt, *Ktl, *Vp, *Vpl, *Vt, *Vtl;   // [B][NH][JP][DH] / [B][NH][DH][JP]
    const uint8_t* valid;                          // [B][JP]
    const float* wth;                              // [NH][NH]
    bf16_t *o, *ol; int ldo;
    bf16_t *P, *Pl, *Pm, *Pml;                     // saved probabilities [B][NH][n][JP]
    float* stats;                                  // optional [B][NH][n][2]: (row max in the log2 domain, 1 / row sum) -- what xattn2_bwd recomputes from
    const bf16_t *dO, *dOl; int lddo;
    bf16_t *dS, *dSl;                              // [B][NH][n][JP]
    bf16_t *dq, *dql; int lddq;
    float* part_th;                                // [grid][NH*NH]
    int B, n, NH, JP, nkb;
    float scale;
};

__device__ __forceinline__ bf16x8 ldfrag16(const bf16_t* p, bool ok) {
    uint4 u = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 ldfrag8x2(const bf16_t* p0, const bf16_t* p1) {
    const uint2 a = *reinterpret_cast<const uint2*>(p0), b = *reinterpret_cast<const uint2*>(p1);
    uint4 u = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
    bf16_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f2bf_hilo(v[e], h[e], l[e]);
    uint4 uh = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
    uint4 ul = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
    hi = __builtin_bit_cast(bf16x8, uh);
    lo = __builtin_bit_cast(bf16x8, ul);
}
__device__ __forceinline__ void store4(bf16_t* hi, bf16_t* lo, const f32x4& v) {
    bf16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f2bf_hilo(v[e], h[e], l[e]);
    *reinterpret_cast<uint2*>(hi) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    if (lo) *reinterpret_cast<uint2*>(lo) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}
__device__ __forceinline__ f32x4 load4(const bf16_t* hi, const bf16_t* lo) {
    const uint2 u = *reinterpret_cast<const uint2*>(hi);
    f32x4 v = {lo_f(u.x), hi_f(u.x), lo_f(u.y), hi_f(u.y)};
    if (lo) {
        const uint2 w = *reinterpret_cast<const uint2*>(lo);
        v[0] += lo_f(w.x); v[1] += hi_f(w.x); v[2] += lo_f(w.y); v[3] += hi_f(w.y);
    }
    return v;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int DH, bool X3, int NKB>      // NKB > 0: compile-time number of 16-key blocks (branch-free unrolled loops)
__global__ __launch_bounds__(512) void xattn_fwd_kernel(XArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    constexpr int NK = NKB > 0 ? NKB : MAXKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* E = reinterpret_cast<float*>(smem);                 // [2][NH][4][64][4] fp32
    __shared__ float wsh[64];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int c = lane & 15, g4 = lane >> 4;
    const int tiles = (a.n + 31) / 32;
    const int b = blockIdx.x / tiles, q0 = (blockIdx.x % tiles) * 32;
    if (threadIdx.x < a.NH * a.NH) wsh[threadIdx.x] = a.wth[threadIdx.x];
    const size_t bh = (size_t)b * a.NH + h;
    const bf16_t* Kp = a.Kp + bh * a.JP * DH;
    const bf16_t* Vt = a.Vt + bh * DH * a.JP;
    const bf16_t* Kpl = X3 ? a.Kpl + bh * a.JP * DH : nullptr;
    const bf16_t* Vtl = X3 ? a.Vtl + bh * DH * a.JP : nullptr;

    // Q^T fragments (MFMA B operand): [k = d][n = query]
    bf16x8 qf[KS][2], qfl[KS][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + c;
        const size_t g = ((size_t)b * a.n + qi) * a.ldq + h * DH + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks][qb] = ldfrag16(a.q + g + ks * 32, qi < a.n);
            if (X3) qfl[ks][qb] = ldfrag16(a.ql + g + ks * 32, qi < a.n);
        }
    }
    // S^T = K Q^T
    f32x4 S[NK][2];
    // K fragments are software-pipelined PF key blocks ahead so their L2 latency overlaps the MFMAs
    constexpr int PF = 3;
    bf16x8 kq[PF][KS], kql[PF][KS];
    auto ldk = [&](int kb, int slot) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t go = (size_t)(kb * 16 + c) * DH + ks * 32 + g4 * 8;
            kq[slot][ks] = ldfrag16(Kp + go, true);
            if (X3) kql[slot][ks] = ldfrag16(Kpl + go, true);
        }
    };
#pragma unroll
    for (int pf = 0; pf < PF; ++pf)
        if (pf < NK && (NKB > 0 || pf < a.nkb)) ldk(pf, pf);
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
        S[kb][0] = S[kb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (NKB > 0 || kb < a.nkb) {
            bf16x8 kf[KS], kfl[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { kf[ks] = kq[kb % PF][ks]; if (X3) kfl[ks] = kql[kb % PF][ks]; }
            if (kb + PF < NK && (NKB > 0 || kb + PF < a.nkb)) ldk(kb + PF, kb % PF);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (X3) {
                        S[kb][qb] = MFMA(kfl[ks], qf[ks][qb], S[kb][qb]);
                        S[kb][qb] = MFMA(kf[ks], qfl[ks][qb], S[kb][qb]);
                    }
                    S[kb][qb] = MFMA(kf[ks], qf[ks][qb], S[kb][qb]);
                }
        }
    }
    // scale, mask, fp32 softmax over keys (per query column)
    float mx[2] = {NEG_MAX, NEG_MAX};
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
        if (NKB > 0 || kb < a.nkb) {
            const uint32_t vm = *reinterpret_cast<const uint32_t*>(a.valid + (size_t)b * a.JP + kb * 16 + g4 * 4);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = ((vm >> (8 * r)) & 0xff) ? S[kb][qb][r] * a.scale : NEG_MAX;
                    S[kb][qb][r] = s;
                    mx[qb] = fmaxf(mx[qb], s);
                }
        }
    }
    float sm[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 16, 64));
        mx[qb] = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64));
    }
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
        if (NKB > 0 || kb < a.nkb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = expf(S[kb][qb][r] - mx[qb]);
                    S[kb][qb][r] = e;
                    sm[qb] += e;
                }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        sm[qb] += __shfl_xor(sm[qb], 16, 64);
        sm[qb] += __shfl_xor(sm[qb], 32, 64);
        sm[qb] = 1.f / sm[qb];
        const int qi = q0 + qb * 16 + c;
        if (a.stats && g4 == 0 && qi < a.n)
            *reinterpret_cast<float2*>(a.stats + (bh * a.n + qi) * 2) = make_float2(mx[qb] * 1.4426950408889634f, sm[qb]);
    }
    // normalise + save P
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
        if (NKB > 0 || kb < a.nkb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                S[kb][qb] = S[kb][qb] * sm[qb];
                const int qi = q0 + qb * 16 + c;
                if (a.P && qi < a.n) {
                    const size_t go = (bh * a.n + qi) * a.JP + kb * 16 + g4 * 4;
                    store4(a.P + go, a.Pl ? a.Pl + go : nullptr, S[kb][qb]);
                }
            }
    __syncthreads();   // wsh visible
    float wrow[8];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) wrow[hh] = hh < a.NH ? wsh[h * a.NH + hh] : 0.f;

    // 32-key chunks: exchange P across heads, mix, O^T += V^T P'^T
    f32x4 O[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db) O[db][0] = O[db][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nch = a.nkb / 2;
#pragma unroll
    for (int ch = 0; ch < NK / 2; ++ch) {
        if (NKB > 0 || ch < nch) {
            float* Eb = E + (size_t)(ch & 1) * a.NH * 4 * 64 * 4;
            // V^T fragments of this chunk: issued before the exchange barrier so they arrive during the mix
            bf16x8 vfr[DB], vfrl[DB];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const size_t go = (size_t)(db * 16 + c) * a.JP + ch * 32 + g4 * 4;
                vfr[db] = ldfrag8x2(Vt + go, Vt + go + 16);
                if (X3) vfrl[db] = ldfrag8x2(Vtl + go, Vtl + go + 16);
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    *reinterpret_cast<f32x4*>(Eb + ((size_t)(h * 4 + k2 * 2 + qb) * 64 + lane) * 4) = S[2 * ch + k2][qb];
            __syncthreads();
            f32x4 pm[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int hh = 0; hh < 8; ++hh)
                        if (hh < a.NH) acc += wrow[hh] * *reinterpret_cast<const f32x4*>(Eb + ((size_t)(hh * 4 + k2 * 2 + qb) * 64 + lane) * 4);
                    pm[k2][qb] = acc;
                    const int qi = q0 + qb * 16 + c;
                    if (a.Pm && qi < a.n) {
                        const size_t go = (bh * a.n + qi) * a.JP + (2 * ch + k2) * 16 + g4 * 4;
                        store4(a.Pm + go, a.Pml ? a.Pml + go : nullptr, acc);
                    }
                }
            // P'^T as MFMA B operand: k-slot (g4, j): j<4 -> key 4g4+j of block 2ch, j>=4 -> key 4g4+j-4 of block 2ch+1
            bf16x8 pf[2], pfl[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float v8[8] = {pm[0][qb][0], pm[0][qb][1], pm[0][qb][2], pm[0][qb][3],
                                     pm[1][qb][0], pm[1][qb][1], pm[1][qb][2], pm[1][qb][3]};
                split8(v8, pf[qb], pfl[qb]);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (X3) {
                        O[db][qb] = MFMA(vfrl[db], pf[qb], O[db][qb]);
                        O[db][qb] = MFMA(vfr[db], pfl[qb], O[db][qb]);
                    }
                    O[db][qb] = MFMA(vfr[db], pf[qb], O[db][qb]);
                }
            }
        }
    }
    // O^T[d = 16db + 4g4 + r][query = 16qb + c]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + c;
        if (qi >= a.n) continue;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const size_t go = ((size_t)b * a.n + qi) * a.ldo + h * DH + db * 16 + g4 * 4;
            store4(a.o + go, a.ol ? a.ol + go : nullptr, O[db][qb]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, query-centric part
// ------------------------------------------------------------------------------------------------
template <int DH, bool X3, int NKB>      // NKB > 0: compile-time number of 16-key blocks (branch-free unrolled loops)
__global__ __launch_bounds__(512) void xattn_bwd_kernel(XArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    constexpr int NK = NKB > 0 ? NKB : MAXKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* E = reinterpret_cast<float*>(smem);                 // [2][NH][4][64][4]
    __shared__ float wsh[64];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int c = lane & 15, g4 = lane >> 4;
    const int tiles = (a.n + 31) / 32;
    const int b = blockIdx.x / tiles, q0 = (blockIdx.x % tiles) * 32;
    if (threadIdx.x < a.NH * a.NH) wsh[threadIdx.x] = a.wth[threadIdx.x];
    const size_t bh = (size_t)b * a.NH + h;
    const bf16_t* Vp = a.Vp + bh * a.JP * DH;
    const bf16_t* Kt = a.Kt + bh * DH * a.JP;
    const bf16_t* Vpl = X3 ? a.Vpl + bh * a.JP * DH : nullptr;
    const bf16_t* Ktl = X3 ? a.Ktl + bh * DH * a.JP : nullptr;

    // dO^T fragments (B operand): [k = d][n = query]   (this wave = head g for the dP' stage)
    bf16x8 df[KS][2], dfl[KS][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + c;
        const size_t g = ((size_t)b * a.n + qi) * a.lddo + h * DH + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            df[ks][qb] = ldfrag16(a.dO + g + ks * 32, qi < a.n);
            if (X3) dfl[ks][qb] = ldfrag16(a.dOl + g + ks * 32, qi < a.n);
        }
    }
    __syncthreads();
    float wcol[8];   // W[g'][h] for this wave's h
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) wcol[gg] = gg < a.NH ? wsh[gg * a.NH + h] : 0.f;

    f32x4 dP[NK][2];
    float dth[8];
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) dth[gg] = 0.f;
    float delta[2] = {0.f, 0.f};
    const int nch = a.nkb / 2;
#pragma unroll
    for (int ch = 0; ch < NK / 2; ++ch) {
        dP[2 * ch][0] = dP[2 * ch][1] = dP[2 * ch + 1][0] = dP[2 * ch + 1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (NKB > 0 || ch < nch) {
            float* Eb = E + (size_t)(ch & 1) * a.NH * 4 * 64 * 4;
            // all global loads of this chunk (V fragments, saved P) are issued up front, ahead of the MFMAs
            // and of the exchange barrier
            bf16x8 vfr[2][KS], vfrl[2][KS];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const size_t go = (size_t)((2 * ch + k2) * 16 + c) * DH + ks * 32 + g4 * 8;
                    vfr[k2][ks] = ldfrag16(Vp + go, true);
                    if (X3) vfrl[k2][ks] = ldfrag16(Vpl + go, true);
                }
            }
            // dP'^T[key][query] = sum_d V[key][d] dO[query][d]   (head = this wave)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        if (X3) {
                            acc[qb] = MFMA(vfrl[k2][ks], df[ks][qb], acc[qb]);
                            acc[qb] = MFMA(vfr[k2][ks], dfl[ks][qb], acc[qb]);
                        }
                        acc[qb] = MFMA(vfr[k2][ks], df[ks][qb], acc[qb]);
                    }
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    *reinterpret_cast<f32x4*>(Eb + ((size_t)(h * 4 + k2 * 2 + qb) * 64 + lane) * 4) = acc[qb];
            }
            // saved P of this chunk: issued before the barrier wait, consumed right after
            uint2 praw[2][2], prawl[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int qi = q0 + qb * 16 + c;
                    praw[k2][qb] = prawl[k2][qb] = make_uint2(0, 0);
                    if (qi < a.n) {
                        const size_t go = (bh * a.n + qi) * a.JP + (2 * ch + k2) * 16 + g4 * 4;
                        praw[k2][qb] = *reinterpret_cast<const uint2*>(a.P + go);
                        if (a.Pl) prawl[k2][qb] = *reinterpret_cast<const uint2*>(a.Pl + go);
                    }
                }
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const uint2 u = praw[k2][qb], ul = prawl[k2][qb];
                    const f32x4 pv = {lo_f(u.x) + lo_f(ul.x), hi_f(u.x) + hi_f(ul.x), lo_f(u.y) + lo_f(ul.y), hi_f(u.y) + hi_f(ul.y)};
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int gg = 0; gg < 8; ++gg)
                        if (gg < a.NH) {
                            const f32x4 e = *reinterpret_cast<const f32x4*>(Eb + ((size_t)(gg * 4 + k2 * 2 + qb) * 64 + lane) * 4);
                            acc += wcol[gg] * e;
                            dth[gg] += (e[0] * pv[0] + e[1] * pv[1]) + (e[2] * pv[2] + e[3] * pv[3]);
                        }
                    dP[2 * ch + k2][qb] = acc;
                    delta[qb] += (acc[0] * pv[0] + acc[1] * pv[1]) + (acc[2] * pv[2] + acc[3] * pv[3]);
                }
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        delta[qb] += __shfl_xor(delta[qb], 16, 64);
        delta[qb] += __shfl_xor(delta[qb], 32, 64);
    }
    // dW_th partial: [g'][h]
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
        const float s = wave_sum(dth[gg]);
        if (lane == 0 && gg < a.NH) a.part_th[(size_t)blockIdx.x * a.NH * a.NH + gg * a.NH + h] = s;
    }
    // ds = P (dP - delta);  dQ^T[d][query] = sum_key K^T[d][key] ds^T[key][query]
    f32x4 dQ[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db) dQ[db][0] = dQ[db][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // explicit 2-deep software pipeline over the 32-key chunks: chunk ch+1's saved P and K^T fragments are
    // requested before chunk ch is processed; a sched_barrier per chunk stops further compiler hoisting
    // (dP is live: register pressure).
    uint2 pr[2][2][2], prl[2][2][2];
    auto ld2 = [&](int ch, int s) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qi = q0 + qb * 16 + c;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const size_t go = (bh * a.n + qi) * a.JP + (2 * ch + k2) * 16 + g4 * 4;
                pr[s][qb][k2] = make_uint2(0, 0);
                if (X3) prl[s][qb][k2] = make_uint2(0, 0);
                if (qi < a.n) {
                    pr[s][qb][k2] = *reinterpret_cast<const uint2*>(a.P + go);
                    if (X3 && a.Pl) prl[s][qb][k2] = *reinterpret_cast<const uint2*>(a.Pl + go);
                }
            }
        }
    };
    if (NKB > 0 || 0 < nch) ld2(0, 0);
#pragma unroll
    for (int ch = 0; ch < NK / 2; ++ch) {
        if (NKB > 0 || ch < nch) {
            const int s = ch & 1;
            if (ch + 1 < NK / 2 && (NKB > 0 || ch + 1 < nch)) ld2(ch + 1, s ^ 1);
            bf16x8 kt[DB], ktl[DB];          // K^T fragments of this chunk (L2-resident): issued now, used after ds
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const size_t go = (size_t)(db * 16 + c) * a.JP + ch * 32 + g4 * 4;
                kt[db] = ldfrag8x2(Kt + go, Kt + go + 16);
                if (X3) ktl[db] = ldfrag8x2(Ktl + go, Ktl + go + 16);
            }
            bf16x8 sf[2], sfl[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int qi = q0 + qb * 16 + c;
                float v8[8];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const uint2 u = pr[s][qb][k2];
                    f32x4 pv = {lo_f(u.x), hi_f(u.x), lo_f(u.y), hi_f(u.y)};
                    if (X3) { const uint2 ul = prl[s][qb][k2]; pv[0] += lo_f(ul.x); pv[1] += hi_f(ul.x); pv[2] += lo_f(ul.y); pv[3] += hi_f(ul.y); }
                    const size_t go = (bh * a.n + qi) * a.JP + (2 * ch + k2) * 16 + g4 * 4;
                    f32x4 dsv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dsv[r] = pv[r] * (dP[2 * ch + k2][qb][r] - delta[qb]);
                        v8[k2 * 4 + r] = dsv[r];
                    }
                    if (qi < a.n) store4(a.dS + go, a.dSl ? a.dSl + go : nullptr, dsv);
                }
                split8(v8, sf[qb], sfl[qb]);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (X3) {
                        dQ[db][qb] = MFMA(ktl[db], sf[qb], dQ[db][qb]);
                        dQ[db][qb] = MFMA(kt[db], sfl[qb], dQ[db][qb]);
                    }
                    dQ[db][qb] = MFMA(kt[db], sf[qb], dQ[db][qb]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + c;
        if (qi >= a.n) continue;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const size_t go = ((size_t)b * a.n + qi) * a.lddq + h * DH + db * 16 + g4 * 4;
            store4(a.dq + go, a.dql ? a.dql + go : nullptr, dQ[db][qb] * a.scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// key/value packing:  kv [B*T, ldkv] (k at col 0, v at col inner) + null_k/null_v -> per (b, h)
//   Kp, Vp [JP][DH] (row 0 = null, rows 1..T = context, rest 0) and their transposes Kt, Vt [DH][JP];
//   valid[b][j] = 1 for j == 0, context_mask[b][j-1] for 1 <= j <= T, 0 beyond.
// ------------------------------------------------------------------------------------------------
// One workgroup = (b, h, block of 64 keys): 16-byte loads of the 64 x DH tile, 16-byte stores of the row-major copies, the tile
// through LDS for the transposes (8 consecutive keys of one channel per 16-byte store).  The first form of this kernel moved 2-byte
// elements with a JP-strided scatter for Kt / Vt and ran at 1.1 TB/s of the 100 MB it touches.
// Any image pointer may be NULL: the image is skipped (the fp16 forward + bf16 backward of the 'bf16x3-fwd' training step read four of the
// eight: K [key][d] and V^T in fp16, K and V [key][d] in bf16 -- half of the 300 MB this kernel wrote per layer was never read).  LO: a
// compile-time flag (the first form indexed its four register images with a run-time part count: 80 B of scratch per lane).
template <bool LO>
__global__ __launch_bounds__(256) void xattn_pack_kernel(const bf16_t* __restrict__ kv, const bf16_t* __restrict__ kvl, int ldkv,
                                                         const float* __restrict__ null_k, const float* __restrict__ null_v,
                                                         const uint8_t* __restrict__ mask, bf16_t* Kp, bf16_t* Kpl, bf16_t* Kt,
                                                         bf16_t* Ktl, bf16_t* Vp, bf16_t* Vpl, bf16_t* Vt, bf16_t* Vtl,
                                                         uint8_t* valid, int B, int T, int NH, int DH, int JP, int lo_f16) {
    // lo_f16: kvl holds the fp16 rendering of the keys / values (not bf16 residuals) and the "lo" images become fp16 images: the
    // null key / value row is then rounded to fp16 as well
    __shared__ bf16_t tile[4][64][72];            // k hi, v hi, k lo, v lo: [key][d], rows padded to 144 bytes (16-byte aligned)
    const int bh = blockIdx.x, b = bh / NH, h = bh % NH, inner = NH * DH, j0 = blockIdx.y * 64;
    if (h == 0 && blockIdx.y == 0)
        for (int j = threadIdx.x; j < JP; j += blockDim.x)
            valid[(size_t)b * JP + j] = j == 0 ? 1 : (j <= T ? (mask ? mask[(size_t)b * T + j - 1] : 1) : 0);
    constexpr int nparts = LO ? 4 : 2;
    const int dchunks = DH / 8;
    // load: (key, 8-channel chunk) per thread and part
    for (int e = threadIdx.x; e < 64 * dchunks; e += blockDim.x) {
        const int jl = e / dchunks, dc = (e % dchunks) * 8, j = j0 + jl;
        uint4 r[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (j == 0) {
            bf16_t hk[8], lk[8], hv[8], lv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f2bf_hilo(null_k[h * DH + dc + t], hk[t], lk[t]); f2bf_hilo(null_v[h * DH + dc + t], hv[t], lv[t]);
                if (lo_f16) { lk[t] = f2h_sat(null_k[h * DH + dc + t]); lv[t] = f2h_sat(null_v[h * DH + dc + t]); }
            }
            r[0] = make_uint4(pack2(hk[0], hk[1]), pack2(hk[2], hk[3]), pack2(hk[4], hk[5]), pack2(hk[6], hk[7]));
            r[1] = make_uint4(pack2(hv[0], hv[1]), pack2(hv[2], hv[3]), pack2(hv[4], hv[5]), pack2(hv[6], hv[7]));
            r[2] = make_uint4(pack2(lk[0], lk[1]), pack2(lk[2], lk[3]), pack2(lk[4], lk[5]), pack2(lk[6], lk[7]));
            r[3] = make_uint4(pack2(lv[0], lv[1]), pack2(lv[2], lv[3]), pack2(lv[4], lv[5]), pack2(lv[6], lv[7]));
        } else if (j <= T) {
            const size_t g = ((size_t)b * T + j - 1) * ldkv + h * DH + dc;
            r[0] = *reinterpret_cast<const uint4*>(kv + g);
            r[1] = *reinterpret_cast<const uint4*>(kv + g + inner);
            if (LO) { r[2] = *reinterpret_cast<const uint4*>(kvl + g); r[3] = *reinterpret_cast<const uint4*>(kvl + g + inner); }
        }
        if (j < JP) {
            const size_t o1 = ((size_t)bh * JP + j) * DH + dc;
            if (Kp) *reinterpret_cast<uint4*>(Kp + o1) = r[0];
            if (Vp) *reinterpret_cast<uint4*>(Vp + o1) = r[1];
            if (LO && Kpl) *reinterpret_cast<uint4*>(Kpl + o1) = r[2];
            if (LO && Vpl) *reinterpret_cast<uint4*>(Vpl + o1) = r[3];
        }
#pragma unroll
        for (int pt = 0; pt < nparts; ++pt) *reinterpret_cast<uint4*>(&tile[pt][jl][dc]) = r[pt];
    }
    __syncthreads();
    // transposes: (channel, 8-key chunk) per thread and part
    for (int e = threadIdx.x; e < DH * 8; e += blockDim.x) {
        const int d = e / 8, jc = (e % 8) * 8, j = j0 + jc;
        if (j >= JP) continue;                      // (JP is a multiple of 32: whole chunks)
        const size_t o2 = ((size_t)bh * DH + d) * JP + j;
#pragma unroll
        for (int pt = 0; pt < nparts; ++pt) {
            bf16_t* dst = pt == 0 ? Kt : (pt == 1 ? Vt : (pt == 2 ? Ktl : Vtl));
            if (!dst) continue;
            bf16_t t8[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) t8[t] = tile[pt][jc + t][d];
            *reinterpret_cast<uint4*>(dst + o2) = make_uint4(pack2(t8[0], t8[1]), pack2(t8[2], t8[3]), pack2(t8[4], t8[5]), pack2(t8[6], t8[7]));
        }
    }
}

// inverse of the packing for the gradients: dKp/dVp fp32 [B][NH][JP][DH] -> dkv bf16 hi[/lo] [B*T, ldkv];
// dnull_k / dnull_v (+)= sum_b row 0
// permuted: the rows of dKp / dVp follow the chunk-permuted key order of the xattn2 / xattn3 backward kernels (their dS / Pm columns):
// key 32 ch + kk sits at row 32 ch + 8 ((kk & 15) >> 2) + 4 (kk >> 4) + (kk & 3); the null key (0) stays at row 0
__device__ __forceinline__ int xattn_key_row(int j, int permuted) {
    if (!permuted) return j;
    const int kk = j & 31;
    return (j & ~31) + 8 * ((kk & 15) >> 2) + 4 * (kk >> 4) + (kk & 3);
}
__global__ __launch_bounds__(256) void xattn_unpack_kernel(const float* __restrict__ dKp, const float* __restrict__ dVp,
                                                           bf16_t* dkv, bf16_t* dkvl, int ldkv, float* dnull_k, float* dnull_v,
                                                           int B, int T, int NH, int DH, int JP, int accumulate, int permuted, int null_last) {
    // null_last (flag bit 2): rows in the key order of amdnuwa_xattn6_bwd -- context key t at position t, the null key at position T
    const int inner = NH * DH;
    const int nrow = xattn_key_row(null_last ? T : 0, permuted);
    if ((int)blockIdx.x == B * NH) {     // null gradients, fixed order over b
        for (int e = threadIdx.x; e < inner; e += blockDim.x) {
            const int h = e / DH, d = e % DH;
            float sk = 0.f, sv = 0.f;
            for (int b = 0; b < B; b += 8) {                     // eight samples in flight, added in order (was one dependent load per sample)
                float vk[8], vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = b + u < B;
                    const size_t o = (((size_t)(in ? b + u : 0) * NH + h) * JP + nrow) * DH + d;
                    const float tk = dKp[o], tv = dVp[o];
                    vk[u] = in ? tk : 0.f; vv[u] = in ? tv : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { sk += vk[u]; sv += vv[u]; }
            }
            dnull_k[e] = accumulate ? dnull_k[e] + sk : sk;
            dnull_v[e] = accumulate ? dnull_v[e] + sv : sv;
        }
        return;
    }
    const int bh = blockIdx.x, b = bh / NH, h = bh % NH, dchunks = DH / 8;
    for (int e = threadIdx.x; e < T * dchunks; e += blockDim.x) {             // 8 channels per thread: 16-byte stores
        const int j = 1 + e / dchunks, dc = (e % dchunks) * 8;
        const size_t o1 = ((size_t)bh * JP + xattn_key_row(null_last ? j - 1 : j, permuted)) * DH + dc;
        const size_t g = ((size_t)b * T + j - 1) * ldkv + h * DH + dc;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float* src = (part ? dVp : dKp) + o1;
            const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
            const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if (!dkvl) {                                         // bf16 mode: the hardware converter (RNE, as f2bf)
                *reinterpret_cast<uint4*>(dkv + g + part * inner) = make_uint4(pack2_rne(f[0], f[1]), pack2_rne(f[2], f[3]), pack2_rne(f[4], f[5]), pack2_rne(f[6], f[7]));
                continue;
            }
            bf16_t hh[8], ll[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) f2bf_hilo(f[t], hh[t], ll[t]);
            *reinterpret_cast<uint4*>(dkv + g + part * inner) = make_uint4(pack2(hh[0], hh[1]), pack2(hh[2], hh[3]), pack2(hh[4], hh[5]), pack2(hh[6], hh[7]));
            *reinterpret_cast<uint4*>(dkvl + g + part * inner) = make_uint4(pack2(ll[0], ll[1]), pack2(ll[2], ll[3]), pack2(ll[4], ll[5]), pack2(ll[6], ll[7]));
        }
    }
}

// dW_th (+)= sum over workgroups of the partials (fixed order)
__global__ __launch_bounds__(1024) void xattn_wth_reduce_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];     // 64 entries (n <= 64) x 16 row groups, fixed combine order
    const int e = threadIdx.x & 63, rg = threadIdx.x >> 6;
    float s = 0.f;
    if (e < n)
        for (int k = rg; k < nblk; k += 16) s += part[(size_t)k * n + e];
    red[rg][e] = s;
    __syncthreads();
    if (rg == 0 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][e];
        out[e] = accumulate ? out[e] + t : t;
    }
}

int check(const amdnuwa_xattn_geom* g) {
    if (!g) return AMDNUWA_ERR_ARG;
    if (g->dim_head != 32 && g->dim_head != 64) return AMDNUWA_ERR_UNSUPPORTED;
    if (g->heads < 1 || g->heads > 8) return AMDNUWA_ERR_UNSUPPORTED;
    if (g->T < 1 || g->T + 1 > MAXKB * 16) return AMDNUWA_ERR_UNSUPPORTED;
    if (g->JP != ((g->T + 1 + 31) / 32) * 32) return AMDNUWA_ERR_ARG;
    return AMDNUWA_OK;
}

}  // namespace

extern "C" int amdnuwa_xattn_jp(int T) { return ((T + 1 + 31) / 32) * 32; }

static int xattn_pack_impl(const amdnuwa_xattn_geom* g, const uint16_t* kv, const uint16_t* kv_lo, int ldkv, const float* null_k,
                           const float* null_v, const uint8_t* context_mask, const amdnuwa_xattn_kv* p, int lo_f16, hipStream_t stream);

extern "C" int amdnuwa_xattn_pack(const amdnuwa_xattn_geom* g, const uint16_t* kv, const uint16_t* kv_lo, int ldkv,
                                  const float* null_k, const float* null_v, const uint8_t* context_mask,
                                  const amdnuwa_xattn_kv* p, hipStream_t stream) {
    return xattn_pack_impl(g, kv, kv_lo, ldkv, null_k, null_v, context_mask, p, 0, stream);
}

extern "C" int amdnuwa_xattn_pack_f16(const amdnuwa_xattn_geom* g, const uint16_t* kv, const uint16_t* kv_f16, int ldkv,
                                      const float* null_k, const float* null_v, const uint8_t* context_mask,
                                      const amdnuwa_xattn_kv* p, hipStream_t stream) {
    if (!kv_f16) return AMDNUWA_ERR_ARG;
    return xattn_pack_impl(g, kv, kv_f16, ldkv, null_k, null_v, context_mask, p, 1, stream);
}

static int xattn_pack_impl(const amdnuwa_xattn_geom* g, const uint16_t* kv, const uint16_t* kv_lo, int ldkv, const float* null_k,
                           const float* null_v, const uint8_t* context_mask, const amdnuwa_xattn_kv* p, int lo_f16, hipStream_t stream) {
    int rc = check(g);
    if (rc) return rc;
    // (every image is optional -- a NULL pointer skips it -- but something must be asked for)
    if (!kv || !null_k || !null_v || !p || !p->valid || ldkv % 8) return AMDNUWA_ERR_ARG;
    if (!p->Kp && !p->Kt && !p->Vp && !p->Vt && !p->Kp_lo && !p->Kt_lo && !p->Vp_lo && !p->Vt_lo) return AMDNUWA_ERR_ARG;
    if (!kv_lo && (p->Kp_lo || p->Kt_lo || p->Vp_lo || p->Vt_lo)) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    const dim3 grid(g->B * g->heads, (g->JP + 63) / 64);
    if (kv_lo)
        hipLaunchKernelGGL(xattn_pack_kernel<true>, grid, dim3(256), 0, stream, kv, kv_lo, ldkv, null_k, null_v, context_mask, p->Kp, p->Kp_lo, p->Kt,
                           p->Kt_lo, p->Vp, p->Vp_lo, p->Vt, p->Vt_lo, p->valid, g->B, g->T, g->heads, g->dim_head, g->JP, lo_f16);
    else
        hipLaunchKernelGGL(xattn_pack_kernel<false>, grid, dim3(256), 0, stream, kv, kv_lo, ldkv, null_k, null_v, context_mask, p->Kp, (bf16_t*)nullptr, p->Kt,
                           (bf16_t*)nullptr, p->Vp, (bf16_t*)nullptr, p->Vt, (bf16_t*)nullptr, p->valid, g->B, g->T, g->heads, g->dim_head, g->JP, lo_f16);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

static XArgs make_args(const amdnuwa_xattn_geom* g, const amdnuwa_xattn_kv* p, const float* w_th) {
    XArgs a{};
    a.Kp = p->Kp; a.Kpl = p->Kp_lo; a.Kt = p->Kt; a.Ktl = p->Kt_lo; a.Vp = p->Vp; a.Vpl = p->Vp_lo; a.Vt = p->Vt; a.Vtl = p->Vt_lo;
    a.valid = p->valid; a.wth = w_th;
    a.B = g->B; a.n = g->n; a.NH = g->heads; a.JP = g->JP; a.nkb = g->JP / 16; a.scale = g->scale;
    return a;
}

extern "C" int amdnuwa_xattn_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq,
                                 const amdnuwa_xattn_kv* p, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo,
                                 uint16_t* P, uint16_t* P_lo, uint16_t* Pm, uint16_t* Pm_lo, hipStream_t stream) {
    return amdnuwa_xattn_fwd_stats(g, q, q_lo, ldq, p, w_th, o, o_lo, ldo, P, P_lo, Pm, Pm_lo, nullptr, stream);
}

extern "C" int amdnuwa_xattn_fwd_stats(const amdnuwa_xattn_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq,
                                       const amdnuwa_xattn_kv* p, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo,
                                       uint16_t* P, uint16_t* P_lo, uint16_t* Pm, uint16_t* Pm_lo, float* stats, hipStream_t stream) {
    int rc = check(g);
    if (rc) return rc;
    if (!q || !p || !w_th || !o || ldq % 8 || ldo % 4) return AMDNUWA_ERR_ARG;
    const bool x3 = q_lo != nullptr;
    if (x3 && (!p->Kp_lo || !p->Vt_lo)) return AMDNUWA_ERR_ARG;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    XArgs a = make_args(g, p, w_th);
    a.q = q; a.ql = q_lo; a.ldq = ldq; a.o = o; a.ol = o_lo; a.ldo = ldo;
    a.P = P; a.Pl = P_lo; a.Pm = Pm; a.Pml = Pm_lo; a.stats = stats;
    const int tiles = (g->n + 31) / 32;
    dim3 grid(g->B * tiles), block(g->heads * 64);
    const size_t lds = (size_t)2 * g->heads * 4 * 64 * 4 * sizeof(float);
#define XF(DH_, X3_, NKB_)                                                                                        \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)xattn_fwd_kernel<DH_, X3_, NKB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((xattn_fwd_kernel<DH_, X3_, NKB_>), grid, block, lds, stream, a);                      \
    } while (0)
    // production shape (dim_head 64, 256 text tokens -> 18 key blocks, bf16): branch-free instantiation
    if (g->dim_head == 64 && !x3 && a.nkb == 18 && g_amdnuwa_tuning[5] == 0) XF(64, false, 18);
    else if (g->dim_head == 64) { if (x3) XF(64, true, 0); else XF(64, false, 0); }
    else { if (x3) XF(32, true, 0); else XF(32, false, 0); }
#undef XF
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_xattn_bwd_workspace_bytes(const amdnuwa_xattn_geom* g) {
    if (check(g)) return 0;
    const size_t tiles = (g->n + 31) / 32;
    return (size_t)g->B * tiles * g->heads * g->heads * sizeof(float);
}

extern "C" int amdnuwa_xattn_bwd(const amdnuwa_xattn_geom* g, const uint16_t* dO, const uint16_t* dO_lo, int lddo,
                                 const amdnuwa_xattn_kv* p, const float* w_th, const uint16_t* P, const uint16_t* P_lo,
                                 uint16_t* dS, uint16_t* dS_lo, uint16_t* dq, uint16_t* dq_lo, int lddq, float* dw_th,
                                 int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    int rc = check(g);
    if (rc) return rc;
    if (!dO || !p || !w_th || !P || !dS || !dq || !dw_th || lddo % 8 || lddq % 4) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_xattn_bwd_workspace_bytes(g)) return AMDNUWA_ERR_WORKSPACE;
    const bool x3 = dO_lo != nullptr;
    if (x3 && (!p->Vp_lo || !p->Kt_lo)) return AMDNUWA_ERR_ARG;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    XArgs a = make_args(g, p, w_th);
    a.dO = dO; a.dOl = dO_lo; a.lddo = lddo;
    a.P = (bf16_t*)P; a.Pl = (bf16_t*)P_lo; a.dS = dS; a.dSl = dS_lo; a.dq = dq; a.dql = dq_lo; a.lddq = lddq;
    a.part_th = (float*)workspace;
    const int tiles = (g->n + 31) / 32;
    dim3 grid(g->B * tiles), block(g->heads * 64);
    const size_t lds = (size_t)2 * g->heads * 4 * 64 * 4 * sizeof(float);
#define XB(DH_, X3_, NKB_)                                                                                        \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)xattn_bwd_kernel<DH_, X3_, NKB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((xattn_bwd_kernel<DH_, X3_, NKB_>), grid, block, lds, stream, a);                      \
    } while (0)
    // (a branch-free NKB = 18 instantiation of the backward spills registers -- dP is live throughout -- so the
    //  backward always uses the guarded form with explicit prefetch)
    if (g->dim_head == 64) { if (x3) XB(64, true, 0); else XB(64, false, 0); }
    else { if (x3) XB(32, true, 0); else XB(32, false, 0); }
#undef XB
    LAUNCH_CHECK();
    hipLaunchKernelGGL(xattn_wth_reduce_kernel, dim3(1), dim3(1024), 0, stream, a.part_th, g->B * tiles, g->heads * g->heads, dw_th, accumulate);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_xattn_unpack(const amdnuwa_xattn_geom* g, const float* dKp, const float* dVp, uint16_t* dkv,
                                    uint16_t* dkv_lo, int ldkv, float* dnull_k, float* dnull_v, int accumulate,
                                    hipStream_t stream) {
    int rc = check(g);
    if (rc) return rc;
    if (!dKp || !dVp || !dkv || !dnull_k || !dnull_v || ldkv % 8) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(xattn_unpack_kernel, dim3(g->B * g->heads + 1), dim3(256), 0, stream, dKp, dVp, dkv, dkv_lo, ldkv, dnull_k, dnull_v,
                       g->B, g->T, g->heads, g->dim_head, g->JP, accumulate & 1, (accumulate >> 1) & 1, (accumulate >> 2) & 1);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

AMDNUWA_SAT_ACCESSOR(xattn)
