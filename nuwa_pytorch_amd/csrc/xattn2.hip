// Text cross-attention core, second design (fast bf16 mode, 8 heads x dim_head 64): np.py:339-378.
//
// The first design (xattn.hip) gives every head its own wave and moves the softmax probabilities of a 32-key chunk
// through LDS so the talking-heads Conv2d(h, h, 1) can mix them -- two workgroup barriers per chunk -- and saves P and P'
// (2 x [B][h][n][JP] bf16) for the backward.  Here ONE WAVE owns 16 queries and ALL 8 heads:
//
//   * S^T[h] = K[h] Q[h]^T (keys x queries) for h = 0..7 lands in the SAME lane positions of 8 accumulators, so the head mix
//     P'[g] = sum_h W[g][h] P[h] is 64 register FMAs per value set -- no cross-wave traffic at all;
//   * the 4 waves of a workgroup (64 queries) share each 32-key chunk of K / V (all heads, 64 KiB) through a double-buffered
//     direct-to-LDS ring (global_load_lds, swizzle on the source address, counted vmcnt + raw s_barrier);
//   * the softmax is two-pass: pass 1 = running (max, sum) per (head, query) from QK^T only, pass 2 recomputes QK^T (MFMA is
//     nearly free here), normalises, mixes, and feeds P'^T straight back as the B operand of O^T = V^T P'^T using the
//     permuted-key trick (C-layout registers = k-slots);
//   * only the statistics (max, 1/sum) are saved: the backward RECOMPUTES P the same way.
//
// Backward (query-centric): pass A = delta[h][q] = sum_j dP[h] P[h], dW_th, and P' (bf16, for dV); pass B = ds = P (dP - delta),
// dq = scale * ds K (K^T fragments via ds_read_b64_tr_b16 out of the same K tile), ds written bf16.  dK / dV stay batched TN
// GEMMs over ds / P' (reduction over the 2560 queries), as in the first design.
#include "common.h"
#include "../../include/amdnuwa.h"

// no implicit a*b+c contraction in this file: pass 1 and pass 2 must round the scaled scores identically (every FMA that
// matters for speed is an explicit fmaf)
#pragma clang fp contract(off)

namespace {

constexpr int NH = 8, DH = 64, KS = 2, DB = 4;
constexpr int TILE = 32 * DH * 2;            // one head's 32-key tile of K, V ([key][d]) or V^T ([d][key]): 4 KiB
constexpr int KT_BYTES = NH * TILE;          // 32 KiB
constexpr int STAGE = 2 * KT_BYTES;          // K + V per chunk
constexpr float NEG_MAX = -3.4028234663852886e38f;

struct X2Args {
    const bf16_t* q; int ldq;
    const bf16_t *Kp, *Vp, *Vt;              // [B][NH][JP][DH] / [B][NH][DH][JP]
    const uint8_t* valid;                    // [B][JP]
    const float* wth;                        // [NH][NH]
    bf16_t* o; int ldo;
    bf16_t* ol;                              // xattn4 fp16-operand form only: the bf16 residual of o (o leaves as a hi + lo pair)
    int ol_f16;                              // ... or, != 0, the FP16 rendering of o (the to_out GEMM's fp16 operand)
    float* stats;                            // [B][NH][n][2] = (row max of the scaled, masked scores; 1 / sum of exp)
    const bf16_t* dO; int lddo;
    bf16_t *dS, *Pm;                         // [B][NH][n][JP], keys of every 32-key chunk in the PERMUTED order the lanes hold them: position
                                             // 8 g4 + e of a chunk = key (e < 4 ? 4 g4 + e : 16 + 4 g4 + e - 4), so that a lane's 8 slots are ONE
                                             // 16-byte store (was two 8-byte stores 32 bytes apart: half the store instructions, 64 contiguous
                                             // bytes per query row).  The batched TN GEMMs reduce over queries and do not care; their outputs
                                             // dKp / dVp carry the same row order and amdnuwa_xattn_unpack undoes it (flag bit 1).
    bf16_t* dq; int lddq;
    float* part_th;                          // [grid][NH*NH]
    int B, n, JP, nch;
    int T;                                   // context keys (the null key is key 0, keys 1..T the context, the rest of JP padding)
    float scale;
    int cm;                                  // xattn3_bwd: dS / Pm chunk-major, [b][h][chunk][n][32] instead of [b][h][n][JP] (1 KiB contiguous per store instruction)
    int nostore;                             // the backward skips its dS / Pm stores (the recomputing key side below replaces them;
                                             // also the probe of tuning key 10 bit 4)
    float* nbd;                              // recomputing backward: [2][B][NH][n] = nb (log2 normaliser) and delta per (head, query),
                                             // written by the query side, read by the key side
    float *dKp, *dVp;                        // key side: [B][NH][JP][DH] fp32
    int flags;                               // key side: bit 0 = plain block order (probe)
};

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef __attribute__((address_space(1))) const void* glb_cvptr;
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// [key][d] tile (128-byte rows): 16-byte chunk gc of row r sits at chunk position gc ^ (r & 7)
__device__ __forceinline__ int kd_off(int h, int row, int gc) { return h * TILE + row * 128 + ((gc ^ (row & 7)) << 4); }
// [d][key] tile (64-byte rows): chunk gc of row d sits at gc ^ ((d >> 2) & 3)
__device__ __forceinline__ int dk_off(int h, int d, int gc) { return h * TILE + d * 64 + ((gc ^ ((d >> 2) & 3)) << 4); }

__device__ __forceinline__ bf16x8 lds16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds8x2(const char* p0, const char* p1) {
    const uint2 a = *reinterpret_cast<const uint2*>(p0), b = *reinterpret_cast<const uint2*>(p1);
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ bf16x8 ldg16(const bf16_t* p, bool ok) {
    return __builtin_bit_cast(bf16x8, ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0));
}
template <bool F16 = false>
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    return __builtin_bit_cast(bf16x8, make_uint4(pack2_t<F16>(v[0], v[1]), pack2_t<F16>(v[2], v[3]), pack2_t<F16>(v[4], v[5]), pack2_t<F16>(v[6], v[7])));
}
// K^T (or any [key][d] tile read transposed): lane (c, g4) gets tile[kb*16 + 4*g4 + j][db*16 + c], j = 0..3, for kb = 0, 1:
// the A operand [row = d][k-slot (g4, j)] of dq^T = K^T ds^T under the permuted-key convention
__device__ __forceinline__ bf16x8 lds_tr(const char* base, int h, int db, int c, int g4) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int col = db * 16 + ((c & 3) << 2);                    // first of the 4 d columns this lane ADDRESSES
    const int r0 = 4 * g4 + (c >> 2), r1 = 16 + r0;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + kd_off(h, r0, col >> 3) + ((col >> 2) & 1) * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + kd_off(h, r1, col >> 3) + ((col >> 2) & 1) * 8));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// one 32-key chunk -> LDS stage `buf`: tile A (always [key][d], from `A`) and tile B ([key][d] when B_KD, else [d][key]).
// 32 + 32 one-KiB DMA pieces, 8 + 8 per wave.
template <bool WITH_B, bool B_KD>
__device__ __forceinline__ void stage_chunk(char* smem, int buf, int ch, int b, int JP, const bf16_t* A, const bf16_t* Bsrc, int wave, int lane) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pi = wave + 4 * i, h = pi >> 2, p = pi & 3;
        const int r = 8 * p + (lane >> 3), gc = (lane & 7) ^ (lane >> 3);
        const bf16_t* src = A + ((size_t)(b * NH + h) * JP + ch * 32 + r) * DH + gc * 8;
        __builtin_amdgcn_global_load_lds((glb_cvptr)src, (lds_vptr)(base + h * TILE + p * 1024), 16, 0, 0);
    }
    if (WITH_B) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pi = wave + 4 * i, h = pi >> 2, p = pi & 3;
            const bf16_t* src;
            if (B_KD) {
                const int r = 8 * p + (lane >> 3), gc = (lane & 7) ^ (lane >> 3);
                src = Bsrc + ((size_t)(b * NH + h) * JP + ch * 32 + r) * DH + gc * 8;
            } else {
                const int d = 16 * p + (lane >> 2), gc = (lane & 3) ^ ((d >> 2) & 3);
                src = Bsrc + ((size_t)(b * NH + h) * DH + d) * JP + ch * 32 + gc * 8;
            }
            __builtin_amdgcn_global_load_lds((glb_cvptr)src, (lds_vptr)(base + KT_BYTES + h * TILE + p * 1024), 16, 0, 0);
        }
    }
}

// S^T chunk of head h: rows = keys (2 blocks of 16), cols = the wave's 16 queries
template <bool F16 = false>
__device__ __forceinline__ void qk_chunk(const char* base, int h, int c, int g4, const bf16x8 (&qf)[KS], f32x4& s0, f32x4& s1) {
    s0 = s1 = f32x4{0.f, 0.f, 0.f, 0.f};
    // all four fragment reads first, then the four MFMAs: one exposed LDS round trip per head instead of two (the compiler keeps the
    // source order: "R R wait M M R R wait M M" in the ISA of the first form)
    bf16x8 k0[KS], k1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        k0[ks] = lds16(base + kd_off(h, c, ks * 4 + g4));
        k1[ks] = lds16(base + kd_off(h, 16 + c, ks * 4 + g4));
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        s0 = mfma16<F16>(k0[ks], qf[ks], s0);
        s1 = mfma16<F16>(k1[ks], qf[ks], s1);
    }
}
// normalised probabilities of the 8 keys this lane holds (slots e = kb*4 + r), 0 where the key is masked:
// P = exp2(s * c1 + nb) with c1 = scale * log2(e) and nb = log2(1 / rowsum) - rowmax2 (statistics kept in the log2 domain)
__device__ __forceinline__ void probs(const f32x4& s0, const f32x4& s1, uint32_t vm0, uint32_t vm1, float c1, float nb, float* P) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (rounded product, then add -- exactly the arithmetic of pass 1, so a row whose only visible key is the null key
        //  reproduces P = 1 and ds = 0 bit for bit instead of 1 + 1e-7)
        P[r] = ((vm0 >> (8 * r)) & 0xff) ? __builtin_amdgcn_exp2f(__fmul_rn(s0[r], c1) + nb) : 0.f;
        P[4 + r] = ((vm1 >> (8 * r)) & 0xff) ? __builtin_amdgcn_exp2f(__fmul_rn(s1[r], c1) + nb) : 0.f;
    }
}
// row g of the head-mix weight (or column h of it, from the transposed copy) out of LDS: 8 broadcast values
struct W8 { float v[8]; };
__device__ __forceinline__ W8 ldw(const float* wsh, int row) {
    const float4 a = *reinterpret_cast<const float4*>(wsh + row * 8), b = *reinterpret_cast<const float4*>(wsh + row * 8 + 4);
    return W8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}

// ------------------------------------------------------------------------------------------------
// The head mix on the matrix pipe (xattn3_* kernels).  P'[g][q, key] = sum_h W[g][h] P[h][q, key] costs 64 FMAs per (q, key) on
// the VALU -- 512 wave instructions per 32-key chunk, the largest single block of the kernel -- although every lane already holds
// the 8 per-head values of its own (query, key) slots.  As an MFMA: B operand (32 x 16) = the lane's 8 head values of ONE slot e
// packed to bf16 (k = 8 * lane_group + h, column = query), A operand (16 x 32) = a constant block pattern of W,
//     A[(kg, gq)][8 kg' + h] = (kg == kg') * W[4 Q + gq][h]         (Q = 0 / 1 selects output heads 0-3 / 4-7),
// so D[(kg, gq)][query] = P'[4 Q + gq] of lane group kg's own slot: output row 4 kg + gq is register gq of lane group kg -- the
// mixed values land in the SAME lanes that own the slot, no data movement.  16 slot-MFMAs per chunk (x 2: W travels as a bf16
// hi + lo pair, so only P itself is rounded, as it is anyway before P'V) instead of 512 FMAs.  The backward mixes dP = W^T dP' the
// same way with the transposed matrix.
// ------------------------------------------------------------------------------------------------
struct MixA { bf16x8 hi[2], lo[2]; };
__device__ __forceinline__ MixA mix_operand(const float* wsrc, int lane) {
    MixA a;
    const int m = lane & 15;
    const bool on = (m >> 2) == (lane >> 4);
#pragma unroll
    for (int Q = 0; Q < 2; ++Q) {
        const W8 w = ldw(wsrc, 4 * Q + (m & 3));
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ph[t] = pack2_rne(w.v[2 * t], w.v[2 * t + 1]);
            pl[t] = pack2_rne(w.v[2 * t] - lo_f(ph[t]), w.v[2 * t + 1] - hi_f(ph[t]));
        }
        a.hi[Q] = __builtin_bit_cast(bf16x8, on ? make_uint4(ph[0], ph[1], ph[2], ph[3]) : make_uint4(0, 0, 0, 0));
        a.lo[Q] = __builtin_bit_cast(bf16x8, on ? make_uint4(pl[0], pl[1], pl[2], pl[3]) : make_uint4(0, 0, 0, 0));
    }
    return a;
}
// the 8 per-head values of slot e, packed as the B operand (element h = head h)
__device__ __forceinline__ bf16x8 pack_heads(const float (&v)[NH][8], int e) {
    return __builtin_bit_cast(bf16x8, make_uint4(pack2_rne(v[0][e], v[1][e]), pack2_rne(v[2][e], v[3][e]),
                                                 pack2_rne(v[4][e], v[5][e]), pack2_rne(v[6][e], v[7][e])));
}
#define MIX(A_, Q_, B_) MFMA((A_).lo[Q_], B_, MFMA((A_).hi[Q_], B_, (f32x4{0.f, 0.f, 0.f, 0.f})))

// ------------------------------------------------------------------------------------------------
// xattn3: the same two-pass structure as xattn2_fwd / xattn2_bwd with every head mix on the matrix pipe
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void xattn3_bwd_kernel(X2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t vsh[96];
    __shared__ float thsh[4][NH * NH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const int tiles = (a.n + 63) / 64;
    const int b = blockIdx.x / tiles, qi = (blockIdx.x % tiles) * 64 + wave * 16 + c;
    const bool qok = qi < a.n;
    if (tid < a.JP / 4) vsh[tid] = reinterpret_cast<const uint32_t*>(a.valid + (size_t)b * a.JP)[tid];
    __shared__ __attribute__((aligned(16))) float wsh[NH * NH], wtsh[NH * NH];          // W[g][h] and its transpose
    if (tid < NH * NH) { const float v = a.wth[tid]; wsh[tid] = v; wtsh[(tid & 7) * 8 + (tid >> 3)] = v; }
    __syncthreads();                                             // (before any DMA is in flight)
    const MixA AW = mix_operand(wsh, lane), AWT = mix_operand(wtsh, lane);
    const float c1 = a.scale * 1.4426950408889634f;
    bf16x8 qf[NH][KS], df[NH][KS];
    float nb[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[h][ks] = ldg16(a.q + ((size_t)b * a.n + qi) * a.ldq + h * DH + ks * 32 + g4 * 8, qok);
            df[h][ks] = ldg16(a.dO + ((size_t)b * a.n + qi) * a.lddo + h * DH + ks * 32 + g4 * 8, qok);
        }
        const float2 st = qok ? *reinterpret_cast<const float2*>(a.stats + (((size_t)b * NH + h) * a.n + qi) * 2) : make_float2(0.f, 1.f);
        nb[h] = qok ? __log2f(st.y) - st.x : 0.f;
    }
    const size_t prow = (size_t)a.n * a.JP;

    // ---- pass A: P' = W P -> Pm;  dP = W^T dP' (both mixes on the matrix pipe);  delta[h] = sum_j dP[h] P[h];
    //      dW_th[g][h] += sum dP'[g] P[h] (the one block that stays on the VALU: its contraction runs over lanes)
    float delta[NH], dth[NH][NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        delta[h] = 0.f;
#pragma unroll
        for (int g = 0; g < NH; ++g) dth[g][h] = 0.f;
    }
    stage_chunk<true, true>(smem, 0, 0, b, a.JP, a.Kp, a.Vp, wave, lane);
    for (int ch = 0; ch < a.nch; ++ch) {
        if (ch + 1 < a.nch) { stage_chunk<true, true>(smem, (ch + 1) & 1, ch + 1, b, a.JP, a.Kp, a.Vp, wave, lane); VMCNT(16); }
        else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        const char* base = smem + (ch & 1) * STAGE;
        const uint32_t vm0 = vsh[ch * 8 + g4], vm1 = vsh[ch * 8 + 4 + g4];
        float P[NH][8];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 s0, s1;
            qk_chunk(base, h, c, g4, qf[h], s0, s1);
            probs(s0, s1, vm0, vm1, c1, nb[h], P[h]);
        }
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            f32x4 D[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) D[e] = MIX(AW, Q, pack_heads(P, e));
            if (qok && !a.nostore && 32 * ch + 4 * g4 <= a.T) {     // (a lane group whose 8 keys are all padding -- 3 of the 4 groups of the last chunk at T = 256 -- writes nothing: the TN GEMMs stop at the last group that holds a key)
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) {
                    bf16_t* dst = a.Pm + ((size_t)b * NH + 4 * Q + rp) * prow + (a.cm ? ((size_t)ch * a.n + qi) * 32 + g4 * 8 : (size_t)qi * a.JP + ch * 32 + g4 * 8);   // (chunk-permuted key order)
                    *reinterpret_cast<uint4*>(dst) = make_uint4(pack2_rne(D[0][rp], D[1][rp]), pack2_rne(D[2][rp], D[3][rp]),
                                                                pack2_rne(D[4][rp], D[5][rp]), pack2_rne(D[6][rp], D[7][rp]));
                }
            }
        }
        // dP'^T[g] = V[g] dO[g]^T, two heads at a time: they go straight into the dW_th sums and, packed, into the B operands of
        // the dP mix (so the full 8 x 8 set never lives in registers)
        uint32_t bw[8][4];
#pragma unroll
        for (int gp = 0; gp < 4; ++gp) {
            float d0[8], d1[8];
            {
                f32x4 s0, s1;
                qk_chunk(base + KT_BYTES, 2 * gp, c, g4, df[2 * gp], s0, s1);
#pragma unroll
                for (int r = 0; r < 4; ++r) { d0[r] = s0[r]; d0[4 + r] = s1[r]; }
                qk_chunk(base + KT_BYTES, 2 * gp + 1, c, g4, df[2 * gp + 1], s0, s1);
#pragma unroll
                for (int r = 0; r < 4; ++r) { d1[r] = s0[r]; d1[4 + r] = s1[r]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bw[e][gp] = pack2_rne(d0[e], d1[e]);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float a0 = dth[2 * gp][h], a1 = dth[2 * gp + 1][h];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0 = fmaf(d0[e], P[h][e], a0); a1 = fmaf(d1[e], P[h][e], a1); }
                dth[2 * gp][h] = a0; dth[2 * gp + 1][h] = a1;
            }
        }
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            f32x4 D[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                D[e] = MIX(AWT, Q, __builtin_bit_cast(bf16x8, make_uint4(bw[e][0], bw[e][1], bw[e][2], bw[e][3])));   // D[e][rp] = dP[4Q + rp]
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                float acc = delta[4 * Q + rp];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf(D[e][rp], P[4 * Q + rp][e], acc);
                delta[4 * Q + rp] = acc;
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    // a query's keys are spread over the 4 lane groups
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        delta[h] += __shfl_xor(delta[h], 16, 64);
        delta[h] += __shfl_xor(delta[h], 32, 64);
        if (a.nbd && g4 == 0 && qok) {                           // what the key side needs per (head, query)
            const size_t at = ((size_t)b * NH + h) * a.n + qi;
            a.nbd[at] = nb[h];
            a.nbd[(size_t)a.B * NH * a.n + at] = delta[h];
        }
    }
    // dW_th partial of this workgroup (fixed order over the 4 waves): lane l ends up with the wave's sum of entry l = g * NH + h
    {
        float tv[NH * NH];
#pragma unroll
        for (int g = 0; g < NH; ++g)
#pragma unroll
            for (int h = 0; h < NH; ++h) tv[g * NH + h] = qok ? dth[g][h] : 0.f;
        thsh[wave][lane] = wave_sum64_transposed(tv, lane);
    }
    __syncthreads();
    if (tid < NH * NH) a.part_th[(size_t)blockIdx.x * NH * NH + tid] = ((thsh[0][tid] + thsh[1][tid]) + thsh[2][tid]) + thsh[3][tid];

    // ---- pass B: ds[h] = P[h] (dP[h] - delta[h]) -> dS;  dq^T[h] += K^T[h] ds^T[h]
    f32x4 dQ[NH][DB];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int db = 0; db < DB; ++db) dQ[h][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage_chunk<true, true>(smem, 0, 0, b, a.JP, a.Kp, a.Vp, wave, lane);
    for (int ch = 0; ch < a.nch; ++ch) {
        if (ch + 1 < a.nch) { stage_chunk<true, true>(smem, (ch + 1) & 1, ch + 1, b, a.JP, a.Kp, a.Vp, wave, lane); VMCNT(16); }
        else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        const char* base = smem + (ch & 1) * STAGE;
        const uint32_t vm0 = vsh[ch * 8 + g4], vm1 = vsh[ch * 8 + 4 + g4];
        bf16x8 bmD[8];
        {
            uint32_t bw[8][4];
#pragma unroll
            for (int gp = 0; gp < 4; ++gp) {
                f32x4 s0, s1, t0, t1;
                qk_chunk(base + KT_BYTES, 2 * gp, c, g4, df[2 * gp], s0, s1);
                qk_chunk(base + KT_BYTES, 2 * gp + 1, c, g4, df[2 * gp + 1], t0, t1);
#pragma unroll
                for (int r = 0; r < 4; ++r) { bw[r][gp] = pack2_rne(s0[r], t0[r]); bw[4 + r][gp] = pack2_rne(s1[r], t1[r]); }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bmD[e] = __builtin_bit_cast(bf16x8, make_uint4(bw[e][0], bw[e][1], bw[e][2], bw[e][3]));
        }
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            f32x4 D[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) D[e] = MIX(AWT, Q, bmD[e]);
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const int h = 4 * Q + rp;
                f32x4 s0, s1;
                float P[8], ds[8];
                qk_chunk(base, h, c, g4, qf[h], s0, s1);
                probs(s0, s1, vm0, vm1, c1, nb[h], P);
#pragma unroll
                for (int e = 0; e < 8; ++e) ds[e] = P[e] * (D[e][rp] - delta[h]);
                const uint2 lo = make_uint2(pack2_rne(ds[0], ds[1]), pack2_rne(ds[2], ds[3]));
                const uint2 hi = make_uint2(pack2_rne(ds[4], ds[5]), pack2_rne(ds[6], ds[7]));
                if (qok && !a.nostore && 32 * ch + 4 * g4 <= a.T) {
                    bf16_t* dst = a.dS + ((size_t)b * NH + h) * prow + (a.cm ? ((size_t)ch * a.n + qi) * 32 + g4 * 8 : (size_t)qi * a.JP + ch * 32 + g4 * 8);  // (chunk-permuted key order)
                    *reinterpret_cast<uint4*>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                const bf16x8 sf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
                for (int db = 0; db < DB; ++db) dQ[h][db] = MFMA(lds_tr(base, h, db, c, g4), sf, dQ[h][db]);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (qok) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                bf16_t* dst = a.dq + ((size_t)b * a.n + qi) * a.lddq + h * DH + db * 16 + g4 * 4;
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack2_rne(dQ[h][db][0] * a.scale, dQ[h][db][1] * a.scale),
                                                            pack2_rne(dQ[h][db][2] * a.scale, dQ[h][db][3] * a.scale));
            }
    }
}

// ------------------------------------------------------------------------------------------------
// xattn4: TWO waves per 16 queries, each owning 4 of the 8 heads (scores, softmax, its 4 output heads' P'V accumulators), so a wave's
// state (64 accumulator + 32 query-fragment registers + working set) fits 256 VGPRs and the workgroup's 8 waves put two waves on every
// SIMD.  The xattn2 / xattn3 kernels run ONE wave per SIMD at ~500 registers and are bound by that single wave's instruction issue
// (PMC r02b: VALU issue 43 % of the wave cycles at ~4.4 cycles per instruction, matrix pipe 10 % busy); two waves per SIMD issue
// alternately.  The head mix needs all 8 heads of a (query, key) slot: after the softmax the pair exchanges its bf16-packed
// probabilities through LDS (4 KiB per wave per chunk) and each wave runs the mix MFMAs for its own 4 output heads.
// LDS: the K / V ring (2 x 64 KiB) + 8 x 4 KiB exchange slots = the whole 160 KiB (the key-valid bytes and W come from global).
// ------------------------------------------------------------------------------------------------
constexpr int XCH = 4096;                               // exchange slot per wave: 8 slots e x 64 lanes x 8 bytes
template <bool WITH_B, bool B_KD>
__device__ __forceinline__ void stage_chunk8(char* smem, int buf, int ch, int b, int JP, const bf16_t* A, const bf16_t* Bsrc, int wave, int lane) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = wave + 8 * i, h = pi >> 2, p = pi & 3;
        const int r = 8 * p + (lane >> 3), gc = (lane & 7) ^ (lane >> 3);
        const bf16_t* src = A + ((size_t)(b * NH + h) * JP + ch * 32 + r) * DH + gc * 8;
        __builtin_amdgcn_global_load_lds((glb_cvptr)src, (lds_vptr)(base + h * TILE + p * 1024), 16, 0, 0);
    }
    if (WITH_B) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pi = wave + 8 * i, h = pi >> 2, p = pi & 3;
            const bf16_t* src;
            if (B_KD) {
                const int r = 8 * p + (lane >> 3), gc = (lane & 7) ^ (lane >> 3);
                src = Bsrc + ((size_t)(b * NH + h) * JP + ch * 32 + r) * DH + gc * 8;
            } else {
                const int d = 16 * p + (lane >> 2), gc = (lane & 3) ^ ((d >> 2) & 3);
                src = Bsrc + ((size_t)(b * NH + h) * DH + d) * JP + ch * 32 + gc * 8;
            }
            __builtin_amdgcn_global_load_lds((glb_cvptr)src, (lds_vptr)(base + KT_BYTES + h * TILE + p * 1024), 16, 0, 0);
        }
    }
}
// A operand of the head-mix MFMA for output heads 4Q .. 4Q+3 (see mix_operand), rows read straight from global memory
struct MixQ { bf16x8 hi, lo; };
template <bool F16 = false>
__device__ __forceinline__ MixQ mix_operand_q(const float* w /* [8][8], row = output head */, int Q, int lane) {
    const int m = lane & 15;
    const bool on = (m >> 2) == (lane >> 4);
    const float* row = w + (4 * Q + (m & 3)) * 8;
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float x0 = row[2 * t], x1 = row[2 * t + 1];
        ph[t] = pack2_t<F16>(x0, x1);
        pl[t] = pack2_t<F16>(x0 - lo_t<F16>(ph[t]), x1 - hi_t<F16>(ph[t]));
    }
    MixQ a;
    a.hi = __builtin_bit_cast(bf16x8, on ? make_uint4(ph[0], ph[1], ph[2], ph[3]) : make_uint4(0, 0, 0, 0));
    a.lo = __builtin_bit_cast(bf16x8, on ? make_uint4(pl[0], pl[1], pl[2], pl[3]) : make_uint4(0, 0, 0, 0));
    return a;
}
#define MIXQ(A_, B_) MFMA((A_).lo, B_, MFMA((A_).hi, B_, (f32x4{0.f, 0.f, 0.f, 0.f})))
template <bool F16>
__device__ __forceinline__ f32x4 mixq(const MixQ& A, const bf16x8& B) {
    return mfma16<F16>(A.lo, B, mfma16<F16>(A.hi, B, f32x4{0.f, 0.f, 0.f, 0.f}));
}
// this wave's half of the B operands (its 4 heads of every slot e) -> its exchange slot; after the workgroup barrier
// xch_full() returns the complete 8-head operand of slot e (own half + the partner wave's)
template <bool F16 = false>
__device__ __forceinline__ void xch_put(char* xch_own, int lane, const float (&v)[4][8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
        *reinterpret_cast<uint2*>(xch_own + e * 512 + lane * 8) = make_uint2(pack2_t<F16>(v[0][e], v[1][e]), pack2_t<F16>(v[2][e], v[3][e]));
}
__device__ __forceinline__ bf16x8 xch_full(const char* xch_own, const char* xch_par, int lane, int hh, int e) {
    const uint2 own = *reinterpret_cast<const uint2*>(xch_own + e * 512 + lane * 8);
    const uint2 par = *reinterpret_cast<const uint2*>(xch_par + e * 512 + lane * 8);
    return __builtin_bit_cast(bf16x8, hh ? make_uint4(par.x, par.y, own.x, own.y) : make_uint4(own.x, own.y, par.x, par.y));
}

// F16: the operands (q, the K / V images) are fp16 and every MFMA of the kernel -- scores, head mix, P'V -- is the fp16 one; the
// probabilities are packed to fp16, W travels as an fp16 hi + lo pair, and o leaves as a bf16 hi + lo pair (a.ol): the forward
// cross-attention core of the 'bf16x3-fwd' precision mode.  Statistics as in the bf16 form (the bf16 backward recomputes from them).
template <bool F16>
__global__ __launch_bounds__(512, 2) void xattn4_fwd_kernel(X2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NHH = NH / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = wave >> 1, hh = wave & 1;                   // query tile of the workgroup, head half (heads 4 hh .. 4 hh + 3)
    const int c = lane & 15, g4 = lane >> 4;
    const int tiles = (a.n + 63) / 64;
    const int b = blockIdx.x / tiles, qi = (blockIdx.x % tiles) * 64 + tile * 16 + c;
    const bool qok = qi < a.n;
    char* xch_own = smem + 2 * STAGE + wave * XCH;
    const char* xch_par = smem + 2 * STAGE + (wave ^ 1) * XCH;
    const uint32_t* vwords = reinterpret_cast<const uint32_t*>(a.valid + (size_t)b * a.JP);
    const MixQ AW = mix_operand_q<F16>(a.wth, hh, lane);
    const float c1 = a.scale * 1.4426950408889634f;
    bf16x8 qf[NHH][KS];
#pragma unroll
    for (int h = 0; h < NHH; ++h)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[h][ks] = ldg16(a.q + ((size_t)b * a.n + qi) * a.ldq + (4 * hh + h) * DH + ks * 32 + g4 * 8, qok);

    // ---- pass 1: running (max, sum of exp) of this wave's 4 heads; only K is staged
    float m[NHH], l[NHH];
#pragma unroll
    for (int h = 0; h < NHH; ++h) { m[h] = NEG_MAX; l[h] = 0.f; }
    stage_chunk8<false, false>(smem, 0, 0, b, a.JP, a.Kp, nullptr, wave, lane);
    if (a.nch > 1) { stage_chunk8<false, false>(smem, 1, 1, b, a.JP, a.Kp, nullptr, wave, lane); VMCNT(4); }
    else VMCNT(0);
    __builtin_amdgcn_s_barrier();                                 // chunk 0 has landed for every wave
    for (int ch = 0; ch < a.nch; ++ch) {
        const char* base = smem + (ch & 1) * STAGE;
        const uint32_t vm0 = vwords[ch * 8 + g4], vm1 = vwords[ch * 8 + 4 + g4];
#pragma unroll
        for (int h = 0; h < NHH; ++h) {
            f32x4 s0, s1;
            qk_chunk<F16>(base, 4 * hh + h, c, g4, qf[h], s0, s1);
            float s[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] = ((vm0 >> (8 * r)) & 0xff) ? s0[r] * c1 : NEG_MAX;
                s[4 + r] = ((vm1 >> (8 * r)) & 0xff) ? s1[r] * c1 : NEG_MAX;
            }
            const float cm = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            const float mn = fmaxf(m[h], cm);
            float acc = l[h] * __builtin_amdgcn_exp2f(m[h] - mn);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += __builtin_amdgcn_exp2f(s[e] - mn);
            l[h] = acc; m[h] = mn;
        }
        // ONE ring barrier per chunk: own pieces of chunk ch + 1 (issued a whole iteration ago) have landed, and after the barrier
        // (a) everyone's have, (b) everyone is done reading stage ch & 1, which chunk ch + 2 may now overwrite
        VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (ch + 2 < a.nch) stage_chunk8<false, false>(smem, ch & 1, ch + 2, b, a.JP, a.Kp, nullptr, wave, lane);
    }
    float nb[NHH];
#pragma unroll
    for (int h = 0; h < NHH; ++h) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float m2 = __shfl_xor(m[h], off, 64), l2 = __shfl_xor(l[h], off, 64);
            const float mn = fmaxf(m[h], m2);
            l[h] = l[h] * __builtin_amdgcn_exp2f(m[h] - mn) + l2 * __builtin_amdgcn_exp2f(m2 - mn);
            m[h] = mn;
        }
        const float il = 1.f / l[h];
        nb[h] = __log2f(il) - m[h];
        if (a.stats && g4 == 0 && qok) *reinterpret_cast<float2*>(a.stats + (((size_t)b * NH + 4 * hh + h) * a.n + qi) * 2) = make_float2(m[h], il);
    }

    // ---- pass 2: P of the own heads, exchange, head mix for the own output heads, O^T[g] += V^T[g] P'^T[g]
    f32x4 O[NHH][DB];
#pragma unroll
    for (int g = 0; g < NHH; ++g)
#pragma unroll
        for (int db = 0; db < DB; ++db) O[g][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage_chunk8<true, false>(smem, 0, 0, b, a.JP, a.Kp, a.Vt, wave, lane);
    if (a.nch > 1) { stage_chunk8<true, false>(smem, 1, 1, b, a.JP, a.Kp, a.Vt, wave, lane); VMCNT(8); }
    else VMCNT(0);
    __builtin_amdgcn_s_barrier();                                 // chunk 0 has landed for every wave
    for (int ch = 0; ch < a.nch; ++ch) {
        const char* base = smem + (ch & 1) * STAGE;
        const uint32_t vm0 = vwords[ch * 8 + g4], vm1 = vwords[ch * 8 + 4 + g4];
        {
            float P[NHH][8];
#pragma unroll
            for (int h = 0; h < NHH; ++h) {
                f32x4 s0, s1;
                qk_chunk<F16>(base, 4 * hh + h, c, g4, qf[h], s0, s1);
                probs(s0, s1, vm0, vm1, c1, nb[h], P[h]);
            }
            xch_put<F16>(xch_own, lane, P);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 D[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            D[e] = mixq<F16>(AW, xch_full(xch_own, xch_par, lane, hh, e));   // D[e][rp] = P'[4 hh + rp] of slot e
        }
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const int g = 4 * hh + rp;
            const float pv[8] = {D[0][rp], D[1][rp], D[2][rp], D[3][rp], D[4][rp], D[5][rp], D[6][rp], D[7][rp]};
            const bf16x8 pf = pack8<F16>(pv);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int d = db * 16 + c;
                const bf16x8 vf = lds8x2(base + KT_BYTES + dk_off(g, d, g4 >> 1) + (g4 & 1) * 8,
                                         base + KT_BYTES + dk_off(g, d, 2 + (g4 >> 1)) + (g4 & 1) * 8);
                O[rp][db] = mfma16<F16>(vf, pf, O[rp][db]);
            }
        }
        // ONE ring barrier per chunk: own pieces of chunk ch + 1 (issued a whole iteration ago) have landed, and after the barrier
        // (a) everyone's have, (b) everyone is done reading stage ch & 1, which chunk ch + 2 may now overwrite
        VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (ch + 2 < a.nch) stage_chunk8<true, false>(smem, ch & 1, ch + 2, b, a.JP, a.Kp, a.Vt, wave, lane);
    }
    if (qok) {
#pragma unroll
        for (int rp = 0; rp < NHH; ++rp)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const size_t go = ((size_t)b * a.n + qi) * a.ldo + (4 * hh + rp) * DH + db * 16 + g4 * 4;
                const uint32_t h01 = pack2_rne(O[rp][db][0], O[rp][db][1]), h23 = pack2_rne(O[rp][db][2], O[rp][db][3]);
                *reinterpret_cast<uint2*>(a.o + go) = make_uint2(h01, h23);
                if (F16 && a.ol)
                    *reinterpret_cast<uint2*>(a.ol + go) = a.ol_f16 ?
                        make_uint2(pack2_f16_sat(O[rp][db][0], O[rp][db][1]), pack2_f16_sat(O[rp][db][2], O[rp][db][3])) :
                        make_uint2(pack2_rne(O[rp][db][0] - lo_f(h01), O[rp][db][1] - hi_f(h01)), pack2_rne(O[rp][db][2] - lo_f(h23), O[rp][db][3] - hi_f(h23)));
            }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, key side (xattn5): dK / dV WITHOUT the dS / P' round trip through HBM.
//
// The query-centric kernel above used to write ds and P' ([B][h][n][JP] bf16 each: 3 GB at b = 128) for two batched TN GEMMs to
// read back -- measured 521 us of stores + 989 us of GEMMs per layer call.  This kernel recomputes both with the roles swapped:
// a workgroup owns 32 KEYS of one sample, its waves hold the K / V fragments of their 16 keys for ALL heads in registers, and the
// queries stream through LDS, 32 at a time (Q and dO rows of all heads = 64 KiB per step, double-buffered LDS-DMA ring, plus the
// 2 KiB of per-(head, query) statistics nb / delta the query side left).  Per step, with lane = (key c, query group g4):
//     S[h]   = Q[h] K[h]^T          A = Q tile rows (LDS), B = K fragment            -> P[h] = exp2(c1 S + nb)   (8 query slots)
//     P'[g]  = sum_h W[g][h] P[h]   the MFMA head mix of xattn3, layout-agnostic     -> B operand of dV^T[g] += dO[g]^T P'[g]
//     dP'[g] = dO[g] V[g]^T ,  dP = W^T dP' ,  ds[h] = P[h] (dP[h] - delta[h])       -> B operand of dK^T[h] += Q[h]^T ds[h]
// The transposed A operands (dO^T, Q^T: rows = d, k-slots = queries) are ds_read_b64_tr_b16 reads of the same tiles.  The 4 waves
// are 2 key blocks x 2 head halves: both halves compute all 8 heads' S and dP' (inputs of the mixes), each mixes and accumulates
// only ITS 4 output heads (128 accumulator registers).  Rounding points are those of the old path (ds, P' -> bf16, fp32 sums).
// ------------------------------------------------------------------------------------------------
constexpr int ST5 = 2 * KT_BYTES + 2048;         // Q tile + dO tile + nb + delta of 32 queries

__device__ __forceinline__ void stage_queries(char* smem, int buf, int st, int b, const X2Args& a, int wave, int lane) {
    char* base = smem + buf * ST5;
    const size_t row0 = (size_t)b * a.n + st * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pi = wave + 4 * i, h = pi >> 2, p = pi & 3;
        const int r = 8 * p + (lane >> 3), gc = (lane & 7) ^ (lane >> 3);
        dma16_asm(a.q + (row0 + r) * a.ldq + h * DH + gc * 8, base + h * TILE + p * 1024);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pi = wave + 4 * i, h = pi >> 2, p = pi & 3;
        const int r = 8 * p + (lane >> 3), gc = (lane & 7) ^ (lane >> 3);
        dma16_asm(a.dO + (row0 + r) * a.lddo + h * DH + gc * 8, base + KT_BYTES + h * TILE + p * 1024);
    }
    if (!(wave & 1)) {                           // [h][32 queries] fp32: lane l brings head l >> 3, queries 4 (l & 7) ..+3
        const int which = wave >> 1;                 // wave 0: nb, wave 2: delta
        const float* src = a.nbd + (which ? (size_t)a.B * NH * a.n : 0) + ((size_t)b * NH + (lane >> 3)) * a.n + st * 32 + (lane & 7) * 4;
        dma16_asm(src, base + 2 * KT_BYTES + which * 1024);
    }
}

#define MIX1(H_, L_, B_) MFMA(L_, B_, MFMA(H_, B_, (f32x4{0.f, 0.f, 0.f, 0.f})))

// one step (32 queries) of the key side for a wave accumulating heads 4 HH .. 4 HH + 3 (compile-time: the wave's P rows and
// accumulators must stay in registers, a run-time head offset sends them to scratch)
// the four A fragments of one head's 32-row tile (rows c / 16 + c, two k-steps)
struct Frag4 { bf16x8 v[4]; };
__device__ __forceinline__ Frag4 tile_frags(const char* base, int h, int c, int g4) {
    Frag4 f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        f.v[2 * ks] = lds16(base + kd_off(h, c, ks * 4 + g4));
        f.v[2 * ks + 1] = lds16(base + kd_off(h, 16 + c, ks * 4 + g4));
    }
    return f;
}
struct Tr4 { bf16x8 v[DB]; };
__device__ __forceinline__ Tr4 tile_tr(const char* base, int h, int c, int g4) {
    Tr4 t;
#pragma unroll
    for (int db = 0; db < DB; ++db) t.v[db] = lds_tr(base, h, db, c, g4);
    return t;
}

template <int HH>
__device__ __forceinline__ void kv_step(const char* base, int c, int g4, float c1, float kbias, const bf16x8 (&kf)[NH][KS],
                                        const bf16x8 (&vf)[NH][KS], const bf16x8& awh, const bf16x8& awl, const bf16x8& awth,
                                        const bf16x8& awtl, f32x4 (&dK)[4][DB], f32x4 (&dV)[4][DB]) {
    // Every LDS read is issued one unit of work (a head, a group of 4 accumulations) ahead of its use: the compiler keeps the source
    // order, and "read, wait, MFMA" per fragment exposed the full LDS latency 128 times per step (one wave per SIMD: nothing else
    // to run meanwhile).
    constexpr int H0 = 4 * HH;
    const float* nbs = reinterpret_cast<const float*>(base + 2 * KT_BYTES);
    const char* dob = base + KT_BYTES;
    float P[NH][8];
    Frag4 fa = tile_frags(base, 0, c, g4);
    Tr4 ta;
    float4 n0 = *reinterpret_cast<const float4*>(nbs + 4 * g4), n1 = *reinterpret_cast<const float4*>(nbs + 16 + 4 * g4);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const Frag4 cur = fa;
        const float nbv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
        if (h + 1 < NH) {
            fa = tile_frags(base, h + 1, c, g4);
            n0 = *reinterpret_cast<const float4*>(nbs + (h + 1) * 32 + 4 * g4); n1 = *reinterpret_cast<const float4*>(nbs + (h + 1) * 32 + 16 + 4 * g4);
        } else ta = tile_tr(dob, H0, c, g4);
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { s0 = MFMA(cur.v[2 * ks], kf[h][ks], s0); s1 = MFMA(cur.v[2 * ks + 1], kf[h][ks], s1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                            // (kbias = -3e38 on a masked key: P = 0)
            P[h][r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c1, nbv[r] + kbias));
            P[h][4 + r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c1, nbv[4 + r] + kbias));
        }
    }
    {   // P' of the wave's heads -> dV^T += dO^T P'
        f32x4 D[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) D[e] = MIX1(awh, awl, pack_heads(P, e));
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const Tr4 cur = ta;
            if (rp + 1 < 4) ta = tile_tr(dob, H0 + rp + 1, c, g4);
            else fa = tile_frags(dob, 0, c, g4);
            const float t[8] = {D[0][rp], D[1][rp], D[2][rp], D[3][rp], D[4][rp], D[5][rp], D[6][rp], D[7][rp]};
            const bf16x8 pf = pack8(t);
#pragma unroll
            for (int db = 0; db < DB; ++db) dV[rp][db] = MFMA(cur.v[db], pf, dV[rp][db]);
        }
    }
    // dP' of all heads, packed per slot (the B operands of the dP mix)
    uint32_t bw[8][4];
#pragma unroll
    for (int gp = 0; gp < 4; ++gp) {
        f32x4 s[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = 2 * gp + u;
            const Frag4 cur = fa;
            if (g + 1 < NH) fa = tile_frags(dob, g + 1, c, g4);
            else ta = tile_tr(base, H0, c, g4);
            s[u][0] = s[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { s[u][0] = MFMA(cur.v[2 * ks], vf[g][ks], s[u][0]); s[u][1] = MFMA(cur.v[2 * ks + 1], vf[g][ks], s[u][1]); }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { bw[r][gp] = pack2_rne(s[0][0][r], s[1][0][r]); bw[4 + r][gp] = pack2_rne(s[0][1][r], s[1][1][r]); }
    }
    {   // dP of the wave's heads -> ds -> dK^T += Q^T ds
        f32x4 D[8];
        const float* dls = nbs + 256;
        float4 l0 = *reinterpret_cast<const float4*>(dls + H0 * 32 + 4 * g4), l1 = *reinterpret_cast<const float4*>(dls + H0 * 32 + 16 + 4 * g4);
#pragma unroll
        for (int e = 0; e < 8; ++e) D[e] = MIX1(awth, awtl, __builtin_bit_cast(bf16x8, make_uint4(bw[e][0], bw[e][1], bw[e][2], bw[e][3])));
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const Tr4 cur = ta;
            const float dl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            if (rp + 1 < 4) {
                ta = tile_tr(base, H0 + rp + 1, c, g4);
                l0 = *reinterpret_cast<const float4*>(dls + (H0 + rp + 1) * 32 + 4 * g4); l1 = *reinterpret_cast<const float4*>(dls + (H0 + rp + 1) * 32 + 16 + 4 * g4);
            }
            float ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ds[e] = P[H0 + rp][e] * (D[e][rp] - dl[e]);
            const bf16x8 sf = pack8(ds);
#pragma unroll
            for (int db = 0; db < DB; ++db) dK[rp][db] = MFMA(cur.v[db], sf, dK[rp][db]);
        }
    }
}

// the whole query loop of one wave (no values merge across the two head halves: a select between them doubles the live registers)
template <int HH>
__device__ __forceinline__ void kv_run(char* smem, const X2Args& a, int b, int j, int wave, int lane, float c1, float kbias,
                                       const bf16x8 (&kf)[NH][KS], const bf16x8 (&vf)[NH][KS], const bf16x8& awh, const bf16x8& awl,
                                       const bf16x8& awth, const bf16x8& awtl) {
    const int c = lane & 15, g4 = lane >> 4;
    f32x4 dK[4][DB], dV[4][DB];
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
#pragma unroll
        for (int db = 0; db < DB; ++db) dK[rp][db] = dV[rp][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nst = a.n / 32;
    stage_queries(smem, 0, 0, b, a, wave, lane);
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) {
            stage_queries(smem, (st + 1) & 1, st + 1, b, a, wave, lane);
            if (HH == 0) VMCNT(17); else VMCNT(16);              // (the HH = 0 waves bring the statistics: one more piece)
        } else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        kv_step<HH>(smem + (st & 1) * ST5, c, g4, c1, kbias, kf, vf, awh, awl, awth, awtl, dK, dV);
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const size_t at = ((size_t)(b * NH + 4 * HH + rp) * a.JP + j) * DH + db * 16 + 4 * g4;
            *reinterpret_cast<float4*>(a.dKp + at) = make_float4(dK[rp][db][0] * a.scale, dK[rp][db][1] * a.scale, dK[rp][db][2] * a.scale, dK[rp][db][3] * a.scale);
            *reinterpret_cast<float4*>(a.dVp + at) = make_float4(dV[rp][db][0], dV[rp][db][1], dV[rp][db][2], dV[rp][db][3]);
        }
}

__global__ __launch_bounds__(256, 1) void xattn5_bwd_kv_kernel(X2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float wsh[NH * NH], wtsh[NH * NH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const int kb = wave >> 1, hh = wave & 1;
    const int nkc = a.JP / 32;
    // the nkc workgroups of a sample stream the SAME Q / dO rows: keep them on one XCD (one L2) -- block ids go round-robin over the 8
    // XCDs, the remap hands each XCD a contiguous slab of logical ids
    const int bid = (a.flags & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / nkc, j = (bid % nkc) * 32 + kb * 16 + c;                       // this lane's key
    if (tid < NH * NH) { const float v = a.wth[tid]; wsh[tid] = v; wtsh[(tid & 7) * 8 + (tid >> 3)] = v; }
    __syncthreads();                                             // (before any DMA is in flight)
    bf16x8 awh, awl, awth, awtl;                                 // mix operands of this wave's 4 output heads
    {
        const MixA AW = mix_operand(wsh, lane), AWT = mix_operand(wtsh, lane);
        awh = hh ? AW.hi[1] : AW.hi[0]; awl = hh ? AW.lo[1] : AW.lo[0];
        awth = hh ? AWT.hi[1] : AWT.hi[0]; awtl = hh ? AWT.lo[1] : AWT.lo[0];
    }
    const float c1 = a.scale * 1.4426950408889634f;
    const float kbias = a.valid[(size_t)b * a.JP + j] ? 0.f : NEG_MAX;
    bf16x8 kf[NH][KS], vf[NH][KS];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[h][ks] = ldg16(a.Kp + ((size_t)(b * NH + h) * a.JP + j) * DH + ks * 32 + g4 * 8, true);
            vf[h][ks] = ldg16(a.Vp + ((size_t)(b * NH + h) * a.JP + j) * DH + ks * 32 + g4 * 8, true);
        }
    // the fragments are complete HERE as far as the compiler's counter model goes: it cannot see the DMA pieces of the ring, and a
    // wait it placed inside the loop for "the k-th oldest load" would drain the prefetched stage instead
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(kf[h][ks]), "v"(vf[h][ks]));
    if (hh) kv_run<1>(smem, a, b, j, wave, lane, c1, kbias, kf, vf, awh, awl, awth, awtl);
    else kv_run<0>(smem, a, b, j, wave, lane, c1, kbias, kf, vf, awh, awl, awth, awtl);
}

int check2(const amdnuwa_xattn_geom* g) {
    if (!g) return AMDNUWA_ERR_ARG;
    if (g->heads != NH || g->dim_head != DH || g->JP % 32 || g->JP > 288 || g->JP < g->T + 1) return AMDNUWA_ERR_UNSUPPORTED;
    return AMDNUWA_OK;
}

}  // namespace

extern "C" int amdnuwa_xattn2_supported(const amdnuwa_xattn_geom* g) { return check2(g) == AMDNUWA_OK; }

extern "C" int amdnuwa_xattn2_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const amdnuwa_xattn_kv* p, const float* w_th,
                                  uint16_t* o, int ldo, float* stats, hipStream_t stream) {
    int rc = check2(g);
    if (rc) return rc;
    if (!q || !p || !p->Kp || !p->Vt || !p->valid || !w_th || !o || ldq % 8 || ldo % 4) return AMDNUWA_ERR_ARG;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    X2Args a{};
    a.q = q; a.ldq = ldq; a.Kp = p->Kp; a.Vp = p->Vp; a.Vt = p->Vt; a.valid = p->valid; a.wth = w_th;
    a.o = o; a.ldo = ldo; a.stats = stats;
    a.B = g->B; a.n = g->n; a.JP = g->JP; a.nch = g->JP / 32; a.T = g->T; a.scale = g->scale;
    const int tiles = (g->n + 63) / 64;
    const int lds4 = 2 * STAGE + 8 * XCH;
    (void)hipFuncSetAttribute((const void*)xattn4_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
    hipLaunchKernelGGL(xattn4_fwd_kernel<false>, dim3(g->B * tiles), dim3(512), lds4, stream, a);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_xattn2_fwd_f16(const amdnuwa_xattn_geom* g, const uint16_t* q_f16, int ldq, const amdnuwa_xattn_kv* p,
                                      const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, float* stats, hipStream_t stream) {
    int rc = check2(g);
    if (rc) return rc;
    if (!q_f16 || !p || !p->Kp_lo || !p->Vt_lo || !p->valid || !w_th || !o || ldq % 8 || ldo % 4) return AMDNUWA_ERR_ARG;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    X2Args a{};
    a.q = q_f16; a.ldq = ldq; a.Kp = p->Kp_lo; a.Vp = p->Vp_lo; a.Vt = p->Vt_lo; a.valid = p->valid; a.wth = w_th;    // the fp16 images
    a.o = o; a.ol = o_lo; a.ldo = ldo; a.stats = stats; a.ol_f16 = (o_lo && o_lo_f16) ? 1 : 0;
    a.B = g->B; a.n = g->n; a.JP = g->JP; a.nch = g->JP / 32; a.T = g->T; a.scale = g->scale;
    const int tiles = (g->n + 63) / 64;
    const int lds4 = 2 * STAGE + 8 * XCH;
    (void)hipFuncSetAttribute((const void*)xattn4_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
    hipLaunchKernelGGL(xattn4_fwd_kernel<true>, dim3(g->B * tiles), dim3(512), lds4, stream, a);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_xattn2_bwd_workspace_bytes(const amdnuwa_xattn_geom* g) {
    if (check2(g)) return 0;
    return (size_t)g->B * ((g->n + 63) / 64) * NH * NH * sizeof(float);
}

extern "C" int amdnuwa_xattn2_bwd_ex(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                                     const amdnuwa_xattn_kv* p, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm,
                                     uint16_t* dq, int lddq, float* part_th, size_t part_bytes, int flags, hipStream_t stream);
extern "C" int amdnuwa_xattn2_bwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                                  const amdnuwa_xattn_kv* p, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm,
                                  uint16_t* dq, int lddq, float* part_th, size_t part_bytes, hipStream_t stream) {
    return amdnuwa_xattn2_bwd_ex(g, q, ldq, dO, lddo, p, w_th, stats, dS, Pm, dq, lddq, part_th, part_bytes, 0, stream);
}
// flags bit 0: dS / Pm chunk-major (see X2Args::cm; what the whole-M batched amdnuwa_gemm_tn reads with a_chunk32)
extern "C" int amdnuwa_xattn2_bwd_ex(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                                     const amdnuwa_xattn_kv* p, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm,
                                     uint16_t* dq, int lddq, float* part_th, size_t part_bytes, int flags, hipStream_t stream) {
    int rc = check2(g);
    if (rc) return rc;
    if (!q || !dO || !p || !p->Kp || !p->Vp || !p->valid || !w_th || !stats || !dS || !Pm || !dq || ldq % 8 || lddo % 8 || lddq % 4)
        return AMDNUWA_ERR_ARG;
    if (!part_th || part_bytes < amdnuwa_xattn2_bwd_workspace_bytes(g)) return AMDNUWA_ERR_WORKSPACE;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    X2Args a{};
    a.q = q; a.ldq = ldq; a.dO = dO; a.lddo = lddo; a.Kp = p->Kp; a.Vp = p->Vp; a.Vt = p->Vt; a.valid = p->valid; a.wth = w_th;
    a.stats = const_cast<float*>(stats); a.dS = dS; a.Pm = Pm; a.dq = dq; a.lddq = lddq; a.part_th = part_th;
    a.B = g->B; a.n = g->n; a.JP = g->JP; a.nch = g->JP / 32; a.T = g->T; a.scale = g->scale;
    const int tiles = (g->n + 63) / 64;
    a.cm = flags & 1;
    (void)hipFuncSetAttribute((const void*)xattn3_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipLaunchKernelGGL(xattn3_bwd_kernel, dim3(g->B * tiles), dim3(256), 2 * STAGE, stream, a);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

// The recomputing backward: query side (dq, dW_th partials, nb / delta) + key side (dKp / dVp), no dS / P' arrays.
extern "C" int amdnuwa_xattn2_bwd_rc_supported(const amdnuwa_xattn_geom* g) { return check2(g) == AMDNUWA_OK && g->n % 32 == 0; }
extern "C" size_t amdnuwa_xattn2_bwd_rc_stats_bytes(const amdnuwa_xattn_geom* g) {
    return amdnuwa_xattn2_bwd_rc_supported(g) ? (size_t)2 * g->B * NH * g->n * sizeof(float) : 0;
}
extern "C" int amdnuwa_xattn2_bwd_rc(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo,
                                     const amdnuwa_xattn_kv* p, const float* w_th, const float* stats, uint16_t* dq, int lddq,
                                     float* part_th, size_t part_bytes, float* nbd, size_t nbd_bytes, float* dKp, float* dVp,
                                     hipStream_t stream) {
    if (!amdnuwa_xattn2_bwd_rc_supported(g)) return g ? AMDNUWA_ERR_UNSUPPORTED : AMDNUWA_ERR_ARG;
    if (!q || !dO || !p || !p->Kp || !p->Vp || !p->valid || !w_th || !stats || !dq || !dKp || !dVp || ldq % 8 || lddo % 8 || lddq % 4)
        return AMDNUWA_ERR_ARG;
    if (!part_th || part_bytes < amdnuwa_xattn2_bwd_workspace_bytes(g) || !nbd || nbd_bytes < amdnuwa_xattn2_bwd_rc_stats_bytes(g))
        return AMDNUWA_ERR_WORKSPACE;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    X2Args a{};
    a.q = q; a.ldq = ldq; a.dO = dO; a.lddo = lddo; a.Kp = p->Kp; a.Vp = p->Vp; a.Vt = p->Vt; a.valid = p->valid; a.wth = w_th;
    a.stats = const_cast<float*>(stats); a.dq = dq; a.lddq = lddq; a.part_th = part_th;
    a.B = g->B; a.n = g->n; a.JP = g->JP; a.nch = g->JP / 32; a.T = g->T; a.scale = g->scale;
    a.nostore = 1; a.nbd = nbd; a.dKp = dKp; a.dVp = dVp; a.flags = 0;
    (void)hipFuncSetAttribute((const void*)xattn3_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipLaunchKernelGGL(xattn3_bwd_kernel, dim3(g->B * ((g->n + 63) / 64)), dim3(256), 2 * STAGE, stream, a);
    LAUNCH_CHECK();
    (void)hipFuncSetAttribute((const void*)xattn5_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST5);
    hipLaunchKernelGGL(xattn5_bwd_kv_kernel, dim3(g->B * (g->JP / 32)), dim3(256), 2 * ST5, stream, a);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

AMDNUWA_SAT_ACCESSOR(xattn2)
