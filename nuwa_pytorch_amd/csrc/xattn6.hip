// Text cross-attention core, third design ("xattn6"): np.py:339-378 for 8 heads x dim_head 64, any context length.
//
// What changed against xattn4 (xattn2.hip), and why (round 6; the ISA of xattn4 showed one exposed LDS round trip per head in the
// score sweep, one per slot in the head mix, and 6 register moves per P'V operand -- the kernel ran at 6 % of the MFMA peak although
// neither the matrix pipe, the VALU nor the LDS was busy):
//   * K / V travel as IMAGES IN LDS ORDER written once per layer by amdnuwa_xattn6_pack: a 32-key chunk of all heads is 32 KiB of K
//     ([head][key][d], bank swizzle baked in) and 32 KiB of V^T ([head][d][key slot], key slots in the order a lane holds its 8
//     probabilities, swizzle baked in): staging is a linear copy (1 KiB DMA pieces, no per-lane address arithmetic) and EVERY MFMA
//     operand is ONE conflict-free ds_read_b128;
//   * the score scale (scale * log2 e) is the multiplier of the ONE fma that feeds each exp2 (no separate multiply; K itself stays as the
//     projection rounded it: pre-scaling the image costs a second rounding of every key -- 7.0e-4 -> 7.8e-4 on the full-depth logits);
//   * the key mask is the C operand of the first score MFMA (0 or MASK_BIAS per key row, built from one 32-bit word per chunk): masking costs no per-element VALU work;
//   * the learned null key is NOT a key row: its score is a 64-long dot product per (query, head), its probability enters the softmax
//     statistics analytically and its value row is a rank-one update of the output -- T = 256 context keys are 8 chunks, not 9;
//   * all LDS reads of a phase are issued before the first MFMA that needs one;
//   * pass 1 (softmax statistics) walks the keys 64 at a time (K only is staged: 4 ring slots of 32 KiB, one barrier per 64 keys).
// The structure is xattn4's otherwise: TWO waves per 16 queries, 4 heads each, probabilities exchanged through LDS for the head mix
// on the matrix pipe (see xattn2.hip), O^T = V^T P'^T with the permuted-key trick, two passes (the head mix after the softmax forbids
// an online rescale).  LDS: 2 x 64 KiB ring + 4 x 8 KiB exchange = 160 KiB.
#include <type_traits>
#include "common.h"
#include "../../include/amdnuwa.h"

// pass 1 and pass 2 must round the scores identically: no implicit contraction
#pragma clang fp contract(off)

namespace {

constexpr int NH = 8, DH = 64, NHH = 4, KS = 2, DB = 4;
constexpr int TILE = 32 * DH * 2;            // one head's 32-key tile: 4 KiB
constexpr int KT = NH * TILE;                // one chunk of K (or of V^T), all heads: 32 KiB
constexpr int STAGE = 2 * KT;                // pass 2: K + V^T of a chunk
constexpr int XT = 8 * 1024;                 // exchange area of one query tile: 8 slots x 64 lanes x 16 bytes
constexpr int LDS_BYTES = 4 * KT + 4 * XT;
constexpr float MASK_BIAS = -400000.f;       // raw (unscaled) score of a masked key: x scale * log2 e = -72 000 in the log2 domain, exp2 of it == 0

struct X6Args {
    const uint16_t* q; int ldq;              // [B*n, ldq] fp16 (F16) or bf16
    const char *K6, *V6;                     // [B][nch][NH][32][64] images (bytes)
    const uint32_t* vbits;                   // [B][nch]: bit j of word ch = key 32 ch + j takes part
    const float *null_k, *null_v;            // [NH][DH]
    const float* wth;                        // [NH][NH]
    uint16_t *o, *ol; int ldo, ol_f16;
    float* stats;                            // [B][NH][n][2] = (reference maximum in the log2 domain, 1 / sum of exp2)
    int B, n, nch;
    float c1;                                // scale * log2(e)
};

// One 1-KiB LDS-DMA piece with a SCALAR base: source = sbase (wave-uniform, SGPR pair) + voff (per-lane byte offset, one VGPR), lane l lands at
// lds + 16 l.  (dma16_asm of common.h takes a per-lane 64-bit address: eight of them per stage call are 16 VGPRs the persistent kernel
// does not have -- spilled, their reloads carried vmcnt(0) waits into the ring loops.)
// (the LDS destination travels as a byte ADDRESS computed from lds_addr_of(smem) once per kernel: a flat -> LDS pointer cast per piece made
//  the compiler emit a null check that it then mis-selected -- "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(lds_vptr_t)p; }
__device__ __forceinline__ void dma16_s(const char* sbase, uint32_t voff, unsigned lds_byte_addr) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// build variant 'x6t' (-DX6_TIMING=1, tools only): s_memtime stamps of workgroup 0's first item, [wave][step][8]
#ifndef X6_TIMING
#define X6_TIMING 0
#endif
#if X6_TIMING
__device__ unsigned long long* g_x6_stamps = nullptr;
#define STAMP(step, k)                                                                                             \
    do {                                                                                                           \
        if (g_x6_stamps && blockIdx.x == 0 && item == (int)gridDim.x) {   /* the SECOND item of workgroup 0 */                                                         \
            const unsigned long long t__ = __builtin_amdgcn_s_memtime();                                           \
            if (lane == 0) g_x6_stamps[(wave * 16 + (step)) * 8 + (k)] = t__;                                      \
        }                                                                                                          \
    } while (0)
#else
#define STAMP(step, k) do { } while (0)
#endif
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// byte offset of the 16-byte piece (8 d values from 8 gc) of key row `row` inside a head's K tile
__host__ __device__ __forceinline__ int k6_off(int h, int row, int gc) { return h * TILE + row * 128 + ((gc ^ (row & 7)) << 4); }
// byte offset of key-slot group j4 (slots 8 j4 .. 8 j4 + 7) of channel d inside a head's V^T tile
__host__ __device__ __forceinline__ int v6_off(int h, int d, int j4) { return h * TILE + d * 64 + ((j4 ^ (((d >> 3) & 1) << 1)) << 4); }
// key (inside the chunk) that sits in slot i of slot group j4: the order in which a lane of the score MFMAs holds its 8 values
__host__ __device__ __forceinline__ int slot_key(int j4, int i) { return i < 4 ? 4 * j4 + i : 16 + 4 * j4 + (i - 4); }

__device__ __forceinline__ bf16x8 lds16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 ldg16(const uint16_t* p, bool ok) {
    return __builtin_bit_cast(bf16x8, ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0));
}
template <bool F16>
__device__ __forceinline__ float h2f(uint16_t v) { return F16 ? (float)__builtin_bit_cast(_Float16, v) : bf2f(v); }

// A operand of the head-mix MFMA for output heads 4Q .. 4Q + 3 (xattn2.hip, mix_operand_q): a constant block pattern of W as a 16-bit
// hi + lo pair, so that only P itself is rounded
struct MixQ { bf16x8 hi, lo; };
template <bool F16>
__device__ __forceinline__ MixQ mix_operand_q(const float* w, int Q, int lane) {
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

// C operands of a chunk's score MFMAs: 0 where the key takes part, MASK_BIAS where it does not (rows 4 g4 + r / 16 + 4 g4 + r)
__device__ __forceinline__ void chunk_bias(uint32_t w, int g4, f32x4& b0, f32x4& b1) {
    const uint32_t u = w >> (4 * g4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b0[r] = ((u >> r) & 1u) ? 0.f : MASK_BIAS;
        b1[r] = ((u >> (16 + r)) & 1u) ? 0.f : MASK_BIAS;
    }
}

// the 4 K fragments of one head (rows c / 16 + c, two k-steps)
// (the lane part of k6_off does not depend on the head or the row block: (16 kb + c) & 7 == c & 7 -- two per-lane offsets, one per k-step,
//  and compile-time immediates for everything else; likewise ONE per-lane offset for V^T: ((16 db + c) >> 3) & 1 == (c >> 3) & 1)
struct KF { bf16x8 v[2][KS]; };
__device__ __forceinline__ KF k_frags(const char* kbase, int h, int ko0, int ko1) {
    KF f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f.v[kb][0] = lds16(kbase + h * TILE + kb * 2048 + ko0);
        f.v[kb][1] = lds16(kbase + h * TILE + kb * 2048 + ko1);
    }
    return f;
}

template <bool F16>
__global__ __launch_bounds__(512, 2) void xattn6_fwd_kernel(X6Args a) {
    // PERSISTENT: one workgroup per CU walks the (sample, 64-query tile) list; the ring prologue of the NEXT item is issued under the last
    // chunks of this one.
    // LDS: four 32-KiB ring slots + the exchange.  Pass 1 (K only): chunk ch in slot SL[ch & 3], SL = {0, 2, 1, 3}, two chunks per step.
    // Pass 2: K of chunk c in slot 2 (c & 1), V^T of chunk c in slot 2 (c & 1) + 1.  Pass 2 is SKEWED by one chunk: iteration c runs the
    // scores + softmax + exchange puts of chunk c next to the head mix + P'V of chunk c - 1 (independent instruction streams of one wave:
    // the round-6 stamps showed 59 % of an un-skewed iteration in waits -- LDS round trips and two barriers in series with every phase).
    // Hazards of iteration c, all closed by two barriers with almost nothing between them:
    //   reads  xb(c-1) [exchange], kf(c) [K slot c&1] ............... then barrier P: every wave holds its xb(c-1) -> puts(c) may overwrite
    //   reads  vf(c-1) [V slot (c-1)&1]; puts(c); DMA K(c+1) -> K slot (c+1)&1 (held K(c-1): read before Q(c-1)), V(c) -> V slot c&1
    //   (held V(c-2): read before Q(c-1)) .......................... then vmcnt(0) + barrier Q: puts and pieces visible, all reads done
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm0 = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = wave >> 1, hh = wave & 1;                   // query tile of the workgroup, head half (heads 4 hh .. 4 hh + 3)
    const int c = lane & 15, g4 = lane >> 4;
    const int tiles = (a.n + 63) / 64, NT = a.B * tiles;
    const int nch = a.nch;
    char* xch = smem + 4 * KT + tile * XT;
    const int H0 = 4 * hh;
    const int ko0 = k6_off(0, c, g4), ko1 = k6_off(0, c, 4 + g4), vo = v6_off(0, c, g4), xo = lane * 16;

    // one 32-KiB image chunk -> ring slot: 32 pieces, 4 per wave
    auto stage = [&](const char* img, int slot, int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 8 * i;
            dma16_s(img + (size_t)ch * KT + piece * 1024, lane * 16, sm0 + slot * KT + piece * 1024);
        }
    };
    auto p1_slot = [](int ch) { return ((ch & 1) << 1) | ((ch >> 1) & 1); };      // SL = {0, 2, 1, 3}
    // the workgroups of a sample stream the same images: the list is cut into one contiguous slab per XCD (workgroup w runs on XCD w % 8
    // and takes items w, w + grid, ...: all of them = w (mod 8) as the grid is a multiple of 8)
    auto sample_of = [&](int item) { return xcd_remap(item, NT) / tiles; };

    MixQ AW = mix_operand_q<F16>(a.wth, hh, lane);
    // (pinned here: left to itself the compiler sinks the W loads to the head of pass 2 and waits for them with vmcnt(0) -- behind
    //  ring pieces that wait would drain)
    asm volatile("" : "+v"(AW.hi), "+v"(AW.lo));

    int item = blockIdx.x;
    if (item < NT) {                                             // pass-1 ring prologue of the first item
        const char* k6 = a.K6 + (size_t)sample_of(item) * nch * KT;
        stage(k6, p1_slot(0), 0); stage(k6, p1_slot(1), 1);
        if (nch > 2) { stage(k6, p1_slot(2), 2); stage(k6, p1_slot(3), 3); }
    }
    for (; item < NT; item += gridDim.x) {
        STAMP(12, 0);
        const int bid = xcd_remap(item, NT);
        const int b = bid / tiles, qi = (bid % tiles) * 64 + tile * 16 + c;
        const bool qok = qi < a.n;
        const char* k6 = a.K6 + (size_t)b * nch * KT;
        const char* v6 = a.V6 + (size_t)b * nch * KT;
        const int nitem = item + gridDim.x;
        const char* k6n = nitem < NT ? a.K6 + (size_t)sample_of(nitem) * nch * KT : nullptr;
        // the key-mask words of the sample (nch <= 64): one vector load here, a v_readlane per chunk later (a load inside the ring loops
        // would be waited for with vmcnt(0), i.e. drain the DMA ring)
        const uint32_t wv = lane < nch ? a.vbits[(size_t)b * nch + lane] : 0u;
        bf16x8 qf[NHH][KS];
#pragma unroll
        for (int h = 0; h < NHH; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[h][ks] = ldg16(a.q + ((size_t)b * a.n + qi) * a.ldq + (H0 + h) * DH + ks * 32 + g4 * 8, qok);

        // ---- the null key: s_null[h] = c1 * q[h] . null_k[h] (fp32), the same value in all four lane groups of a query
        float sn[NHH];
#pragma unroll
        for (int h = 0; h < NHH; ++h) {
            float acc = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // (rounded to the operand type of the mode, as a key row of the images would be: in the bf16 mode the recomputing backward,
                //  whose images hold the bf16 null key, then sees the very same products -- a sample with no visible context key gets
                //  P_null = 1 and ds = 0 instead of 1 +- 2^-9)
                float4 k0 = *reinterpret_cast<const float4*>(a.null_k + (H0 + h) * DH + ks * 32 + g4 * 8);
                float4 k1 = *reinterpret_cast<const float4*>(a.null_k + (H0 + h) * DH + ks * 32 + g4 * 8 + 4);
                auto r16 = [](float x) { return h2f<F16>(F16 ? f2h_sat(x) : f2bf(x)); };
                k0 = make_float4(r16(k0.x), r16(k0.y), r16(k0.z), r16(k0.w)); k1 = make_float4(r16(k1.x), r16(k1.y), r16(k1.z), r16(k1.w));
                const uint4 u = __builtin_bit_cast(uint4, qf[h][ks]);
                acc = fmaf(h2f<F16>((uint16_t)(u.x & 0xffff)), k0.x, acc); acc = fmaf(h2f<F16>((uint16_t)(u.x >> 16)), k0.y, acc);
                acc = fmaf(h2f<F16>((uint16_t)(u.y & 0xffff)), k0.z, acc); acc = fmaf(h2f<F16>((uint16_t)(u.y >> 16)), k0.w, acc);
                acc = fmaf(h2f<F16>((uint16_t)(u.z & 0xffff)), k1.x, acc); acc = fmaf(h2f<F16>((uint16_t)(u.z >> 16)), k1.y, acc);
                acc = fmaf(h2f<F16>((uint16_t)(u.w & 0xffff)), k1.z, acc); acc = fmaf(h2f<F16>((uint16_t)(u.w >> 16)), k1.w, acc);
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            sn[h] = acc * a.c1;
        }

        // ---- pass 1: running (reference maximum, sum of exp2) of this wave's 4 heads, 64 keys per ring step; the null key opens the
        // sums (counted once: in lane group 0)
        float m[NHH], l[NHH];
#pragma unroll
        for (int h = 0; h < NHH; ++h) { m[h] = sn[h]; l[h] = g4 == 0 ? 1.f : 0.f; }
        STAMP(12, 1);
        VMCNT(0);                                                 // the ring prologue (issued long ago, or just now for the first item)
        STAMP(12, 2);
        __builtin_amdgcn_s_barrier();
        STAMP(12, 3);
        bool k0_staged = false;                                   // chunk 0 of pass 2 already on its way (wave-uniform)
        for (int p = 0; 2 * p < nch; ++p) {
            const int sl0 = p1_slot(2 * p), sl1 = p1_slot(2 * p + 1);
            STAMP(p, 0);
            const char* kb0 = smem + sl0 * KT;
            const char* kb1 = smem + sl1 * KT;
            f32x4 ba0, ba1, bb0, bb1;
            chunk_bias(__builtin_amdgcn_readlane(wv, 2 * p), g4, ba0, ba1);
            chunk_bias(__builtin_amdgcn_readlane(wv, 2 * p + 1), g4, bb0, bb1);
#pragma unroll
            for (int h = 0; h < NHH; ++h) {
                const KF fa = k_frags(kb0, H0 + h, ko0, ko1), fb = k_frags(kb1, H0 + h, ko0, ko1);
                f32x4 s0v = ba0, s1v = ba1, t0 = bb0, t1 = bb1;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s0v = mfma16<F16>(fa.v[0][ks], qf[h][ks], s0v);
                    s1v = mfma16<F16>(fa.v[1][ks], qf[h][ks], s1v);
                    t0 = mfma16<F16>(fb.v[0][ks], qf[h][ks], t0);
                    t1 = mfma16<F16>(fb.v[1][ks], qf[h][ks], t1);
                }
                const float ca = fmaxf(fmaxf(fmaxf(s0v[0], s0v[1]), fmaxf(s0v[2], s0v[3])), fmaxf(fmaxf(s1v[0], s1v[1]), fmaxf(s1v[2], s1v[3])));
                const float cb = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])), fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3])));
                const float mn = fmaxf(m[h], fmaxf(ca, cb) * a.c1);       // (c1 > 0: the maximum of the scaled scores)
                float acc = l[h] * __builtin_amdgcn_exp2f(m[h] - mn);
                float acc2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc += __builtin_amdgcn_exp2f(fmaf(s0v[r], a.c1, -mn));
                    acc2 += __builtin_amdgcn_exp2f(fmaf(s1v[r], a.c1, -mn));
                    acc += __builtin_amdgcn_exp2f(fmaf(t0[r], a.c1, -mn));
                    acc2 += __builtin_amdgcn_exp2f(fmaf(t1[r], a.c1, -mn));
                }
                l[h] = acc + acc2; m[h] = mn;
            }
            STAMP(p, 1);
            // ONE ring barrier per step: own pieces of the next pair (issued a whole step ago) have landed, and after the barrier
            // (a) everyone's have, (b) everyone is done reading this pair's slots, which may now be overwritten: by pair p + 2, or -- if
            // the pair sat in the K slots of pass 2 and is the last but one -- by K of chunk 0 of pass 2
            VMCNT(0);
            STAMP(p, 2);
            __builtin_amdgcn_s_barrier();
            STAMP(p, 3);
            if (2 * p + 4 < nch) { stage(k6, sl0, 2 * p + 4); stage(k6, sl1, 2 * p + 5); }
            else if (2 * p + 2 < nch && sl0 == 0) { stage(k6, 0, 0); k0_staged = true; }
        }
        if (!k0_staged) stage(k6, 0, 0);
        STAMP(12, 4);

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
            const float il = __builtin_amdgcn_rcpf(l[h]);
            nb[h] = -__log2f(l[h]) - m[h];
            if (a.stats && g4 == 0 && qok) *reinterpret_cast<float2*>(a.stats + (((size_t)b * NH + H0 + h) * a.n + qi) * 2) = make_float2(m[h], il);
        }

        // ---- the null key's share first: P_null of all 8 heads through LDS (1 KiB per tile inside ring slot 3: free from the end of pass 1
        // to the top of pass-2 iteration 1), one head mix, and O starts as the rank-one term v_null x P'_null instead of zero
        char* nx = smem + 3 * KT + tile * 1024;
        {
            float pn[NHH];
#pragma unroll
            for (int h = 0; h < NHH; ++h) pn[h] = __builtin_amdgcn_exp2f(sn[h] + nb[h]);
            *reinterpret_cast<uint2*>(nx + lane * 16 + hh * 8) = make_uint2(pack2_t<F16>(pn[0], pn[1]), pack2_t<F16>(pn[2], pn[3]));
        }
        STAMP(12, 5);
        LGKM0();
        VMCNT(0);
        __builtin_amdgcn_s_barrier();                             // K of chunk 0 has landed for every wave; the partner's P_null is there
        STAMP(12, 6);
        // ---- pass 2 (skewed): iteration cc = scores / softmax / puts of chunk cc  +  head mix / P'V of chunk cc - 1
        f32x4 O[NHH][DB];
#pragma unroll
        for (int g = 0; g < NHH; ++g)
#pragma unroll
            for (int db = 0; db < DB; ++db) O[g][db] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 Dn;                                                 // Dn[rp] = P'_null[4 hh + rp] of this lane's query (all lane groups alike)
        {
            const bf16x8 xn = lds16(nx + lane * 16);
            Dn = mfma16<F16>(AW.hi, xn, f32x4{0.f, 0.f, 0.f, 0.f});
            Dn = mfma16<F16>(AW.lo, xn, Dn);
        }
        auto body = [&](auto FIRST_, auto LAST_, int cc) {
            constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
            STAMP(4 + (cc > 11 ? 11 : cc), 0);
            // the pieces of the next iteration: K of chunk cc + 1, V^T of chunk cc (LAST: the first two pass-1 chunks of the next item go
            // into the K slots -- both free)
            if (!LAST) {
                if (cc + 1 < nch) stage(k6, 2 * ((cc + 1) & 1), cc + 1);
                stage(v6, 2 * (cc & 1) + 1, cc);
            } else if (k6n) { stage(k6n, p1_slot(0), 0); stage(k6n, p1_slot(1), 1); }
            bf16x8 xb[8];
            KF kf[NHH];
            if (!FIRST) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[e] = lds16(xch + e * 1024 + xo);
            }
            if (!LAST) {
                const char* kbase = smem + 2 * (cc & 1) * KT;
#pragma unroll
                for (int h = 0; h < NHH; ++h) kf[h] = k_frags(kbase, H0 + h, ko0, ko1);
            }
            if (!FIRST) {
                LGKM0();
                STAMP(4 + (cc > 11 ? 11 : cc), 1);
                __builtin_amdgcn_s_barrier();                     // (P) every wave holds its xb: the exchange may be overwritten
                STAMP(4 + (cc > 11 ? 11 : cc), 2);
            }
            f32x4 D[8];
            if (!FIRST) {
#pragma unroll
                for (int e = 0; e < 8; ++e) D[e] = mfma16<F16>(AW.hi, xb[e], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int e = 0; e < 8; ++e) D[e] = mfma16<F16>(AW.lo, xb[e], D[e]);   // D[e][rp] = P'[4 hh + rp] of slot e (chunk cc - 1)
            }
            f32x4 s0v[NHH], s1v[NHH];
            if (!LAST) {
                f32x4 b0, b1;
                chunk_bias(__builtin_amdgcn_readlane(wv, cc), g4, b0, b1);
#pragma unroll
                for (int h = 0; h < NHH; ++h) {
                    s0v[h] = b0; s1v[h] = b1;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        s0v[h] = mfma16<F16>(kf[h].v[0][ks], qf[h][ks], s0v[h]);
                        s1v[h] = mfma16<F16>(kf[h].v[1][ks], qf[h][ks], s1v[h]);
                    }
                }
            }
            bf16x8 vf[NHH][DB];
            if (!FIRST) {
                const char* vbase = smem + (2 * ((cc - 1) & 1) + 1) * KT;
#pragma unroll
                for (int g = 0; g < NHH; ++g)
#pragma unroll
                    for (int db = 0; db < DB; ++db) vf[g][db] = lds16(vbase + (H0 + g) * TILE + db * 1024 + vo);
            }
            if (!LAST) {
                // probabilities of the own 4 heads -> exchange (slot e: 16 bytes per lane = heads 0..7, own half at byte 8 hh)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float pe[NHH];
#pragma unroll
                    for (int h = 0; h < NHH; ++h) pe[h] = __builtin_amdgcn_exp2f(fmaf(e < 4 ? s0v[h][e & 3] : s1v[h][e & 3], a.c1, nb[h]));
                    *reinterpret_cast<uint2*>(xch + e * 1024 + xo + hh * 8) = make_uint2(pack2_t<F16>(pe[0], pe[1]), pack2_t<F16>(pe[2], pe[3]));
                }
            }
            if (!FIRST) {
#pragma unroll
                for (int rp = 0; rp < NHH; ++rp) {
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pack2_t<F16>(D[0][rp], D[1][rp]), pack2_t<F16>(D[2][rp], D[3][rp]),
                                                                          pack2_t<F16>(D[4][rp], D[5][rp]), pack2_t<F16>(D[6][rp], D[7][rp])));
#pragma unroll
                    for (int db = 0; db < DB; ++db) O[rp][db] = mfma16<F16>(vf[rp][db], pf, O[rp][db]);
                }
            }
            STAMP(4 + (cc > 11 ? 11 : cc), 3);
            LGKM0();
            VMCNT(0);
            STAMP(4 + (cc > 11 ? 11 : cc), 4);
            __builtin_amdgcn_s_barrier();                         // (Q)
            STAMP(4 + (cc > 11 ? 11 : cc), 5);
        };
        {
            body(std::true_type{}, std::false_type{}, 0);
            for (int cc = 1; cc < nch; ++cc) body(std::false_type{}, std::false_type{}, cc);
            // (the null value rows are asked for before the last body: they arrive under it)
            float4 vn[NHH][DB];
#pragma unroll
            for (int rp = 0; rp < NHH; ++rp)
#pragma unroll
                for (int db = 0; db < DB; ++db) vn[rp][db] = *reinterpret_cast<const float4*>(a.null_v + (H0 + rp) * DH + db * 16 + g4 * 4);
            body(std::false_type{}, std::true_type{}, nch);
#pragma unroll
            for (int rp = 0; rp < NHH; ++rp)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    O[rp][db][0] = fmaf(vn[rp][db].x, Dn[rp], O[rp][db][0]); O[rp][db][1] = fmaf(vn[rp][db].y, Dn[rp], O[rp][db][1]);
                    O[rp][db][2] = fmaf(vn[rp][db].z, Dn[rp], O[rp][db][2]); O[rp][db][3] = fmaf(vn[rp][db].w, Dn[rp], O[rp][db][3]);
                }
        }
        if (k6n && nch > 2) { stage(k6n, p1_slot(2), 2); stage(k6n, p1_slot(3), 3); }

        STAMP(13, 0);
        STAMP(13, 1);
        // Output rows.  A lane holds 4 consecutive channels (8 bytes) per (head, 16-channel block); v_permlane16_swap pairs the blocks
        // (db, db + 1) so that every lane owns 8 consecutive channels = ONE 16-byte store (lane group g4: channels 16 (g4 & 1) + 8 (g4 >> 1) ..
        // of the 32-channel pair): half the store instructions for the same bytes (the tail was store-issue-bound: 10 k cycles per item)
        if (true) {
            const int dlane = 16 * (g4 & 1) + 8 * (g4 >> 1);
            float amax = 0.f;
#pragma unroll
            for (int rp = 0; rp < NHH; ++rp)
#pragma unroll
                for (int dp = 0; dp < DB; dp += 2) {
                    const size_t go = ((size_t)b * a.n + qi) * a.ldo + (H0 + rp) * DH + dp * 16 + dlane;
                    uint32_t x0 = pack2_rne(O[rp][dp][0], O[rp][dp][1]), x1 = pack2_rne(O[rp][dp][2], O[rp][dp][3]);
                    uint32_t y0 = pack2_rne(O[rp][dp + 1][0], O[rp][dp + 1][1]), y1 = pack2_rne(O[rp][dp + 1][2], O[rp][dp + 1][3]);
                    uint32_t lx0, lx1, ly0, ly1;                  // the second output: fp16 copy or bf16 residual
                    if (a.ol_f16) {                               // (saturating AND counted: amdnuwa_f16_sat_count)
                        lx0 = pack2_f16_sat_n(O[rp][dp][0], O[rp][dp][1], amax); lx1 = pack2_f16_sat_n(O[rp][dp][2], O[rp][dp][3], amax);
                        ly0 = pack2_f16_sat_n(O[rp][dp + 1][0], O[rp][dp + 1][1], amax); ly1 = pack2_f16_sat_n(O[rp][dp + 1][2], O[rp][dp + 1][3], amax);
                    } else {
                        lx0 = pack2_rne(O[rp][dp][0] - lo_f(x0), O[rp][dp][1] - hi_f(x0)); lx1 = pack2_rne(O[rp][dp][2] - lo_f(x1), O[rp][dp][3] - hi_f(x1));
                        ly0 = pack2_rne(O[rp][dp + 1][0] - lo_f(y0), O[rp][dp + 1][1] - hi_f(y0)); ly1 = pack2_rne(O[rp][dp + 1][2] - lo_f(y1), O[rp][dp + 1][3] - hi_f(y1));
                    }
                    auto swap = [](uint32_t& x, uint32_t& y) {
                        const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
                        x = r[0]; y = r[1];
                    };
                    if (a.o) {                                    // (o == NULL: the fp16 copy alone -- the fp16-gradient backward reads nothing else)
                        swap(x0, y0); swap(x1, y1);
                        if (qok) *reinterpret_cast<uint4*>(a.o + go) = make_uint4(x0, x1, y0, y1);
                    }
                    if (a.ol) {
                        swap(lx0, ly0); swap(lx1, ly1);
                        if (qok) *reinterpret_cast<uint4*>(a.ol + go) = make_uint4(lx0, lx1, ly0, ly1);
                    }
                }
            if (qok) f16_sat_commit(amax);
        }
        STAMP(13, 2);
    }
}

// ------------------------------------------------------------------------------------------------
// images: kv16 [B*T, ldkv] (keys in columns [0, 512), values in [512, 1024), fp16 or bf16) -> K6 / V6 / vbits.
// One workgroup per (sample, chunk).  Chunks beyond the context (nch is even) are all-masked zeros.
// ------------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void xattn6_pack_kernel(const uint16_t* __restrict__ kv, int ldkv, const uint8_t* __restrict__ mask,
                                                          char* __restrict__ K6, char* __restrict__ V6, uint32_t* __restrict__ vbits,
                                                          int T, int nch) {
    __shared__ __attribute__((aligned(16))) uint16_t vt[32][NH * DH + 8];   // the chunk's value rows (row pitch 1040 bytes)
    const int b = blockIdx.x / nch, ch = blockIdx.x % nch, tid = threadIdx.x;
    const size_t cbase = ((size_t)b * nch + ch) * KT;
    if (tid < 64) {
        const int j = 32 * ch + tid;
        const bool ok = tid < 32 && j < T && (mask ? mask[(size_t)b * T + j] != 0 : true);
        const unsigned long long bal = __ballot(ok);
        if (tid == 0) vbits[(size_t)b * nch + ch] = (uint32_t)bal;
    }
    // value rows -> LDS (16-byte pieces: 32 rows x 64 pieces)
    for (int e = tid; e < 32 * 64; e += 256) {
        const int row = e >> 6, pc = e & 63, j = 32 * ch + row;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (j < T) r = *reinterpret_cast<const uint4*>(kv + ((size_t)b * T + j) * ldkv + NH * DH + pc * 8);
        *reinterpret_cast<uint4*>(&vt[row][pc * 8]) = r;
    }
    // K image: piece (h, row, pos) holds the 8 d values of chunk gc = pos ^ (row & 7)
    for (int e = tid; e < NH * 32 * 8; e += 256) {
        const int h = e >> 8, row = (e >> 3) & 31, pos = e & 7, gc = pos ^ (row & 7), j = 32 * ch + row;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (j < T) {
            r = *reinterpret_cast<const uint4*>(kv + ((size_t)b * T + j) * ldkv + h * DH + gc * 8);
        }
        *reinterpret_cast<uint4*>(K6 + cbase + h * TILE + row * 128 + pos * 16) = r;
    }
    __syncthreads();
    // V^T image: piece (h, d, pos) holds key slots 8 j4 .. 8 j4 + 7 of channel d, j4 = pos ^ swizzle(d)
    for (int e = tid; e < NH * 64 * 4; e += 256) {
        const int h = e >> 8, d = (e >> 2) & 63, pos = e & 3, j4 = pos ^ (((d >> 3) & 1) << 1);
        uint16_t t8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t8[i] = vt[slot_key(j4, i)][h * DH + d];
        *reinterpret_cast<uint4*>(V6 + cbase + h * TILE + d * 64 + pos * 16) =
            make_uint4(pack2(t8[0], t8[1]), pack2(t8[2], t8[3]), pack2(t8[4], t8[5]), pack2(t8[6], t8[7]));
    }
}

// ------------------------------------------------------------------------------------------------
// backward, query side (xattn6_bwd): the two-pass recomputing backward of xattn3_bwd (xattn2.hip) -- one wave = 16 queries x all 8
// heads, both head mixes on the matrix pipe, dS / Pm written chunk-major for the batched dK / dV products -- on the forward's recipe:
// bf16 K and V images in LDS order (linear 1-KiB DMA pieces with a scalar base: the per-lane address arithmetic of 16 pieces per chunk
// and wave is gone), the key mask as the C operand of the score MFMAs (no per-element compare / select), the score scale folded into
// ONE fma per probability, no run-time probe branches inside the chunk loops, XCD-aware item order.  Key order of the images and of dS / Pm:
// positions 0..T-1 = the context keys (chunk-aligned), position T = the null key (amdnuwa_xattn_unpack flag bit 2 reads dKp / dVp that way).
// With T % 32 == 0 the null key would be alone in the last chunk: that chunk is no matrix iteration but a rank-one term per pass.
// ------------------------------------------------------------------------------------------------
// a value rounded as an image row of the backward holds it: bf16, or (F16) fp16
template <bool F16> __device__ __forceinline__ float rnd16(float v) { return F16 ? (float)(_Float16)v : bf2f(f2bf(v)); }
struct X6BArgs {
    const uint16_t* q; int ldq;
    const uint16_t* dO; int lddo;
    const char *K6, *V6;                     // [B][nch][NH][32][64] bf16, [key][d] tiles in k6_off order (K unscaled); keys 0..T-1 the context, key T the null key
    const uint32_t* vbits;                   // [B][nch]
    const float *null_k, *null_v;            // [NH][DH]
    const float* wth;
    const float* stats;                      // [B][NH][n][2]
    uint16_t *dS, *Pm;                       // [B][NH][nch][n][32] (chunk-major, chunk-permuted slots)
    uint16_t* dq; int lddq;
    float* part_th;                          // [grid][NH*NH]
    int B, n, nch, T;
    float scale;
};
constexpr float MASK_BIAS_RAW = -400000.f;   // unscaled score of a masked key (x scale * log2 e = -72 000 in the log2 domain)
// Range of the fp16-gradient form.  dO arrives as fp16(S dO) with S chosen from the residual-stream gradient at the top of the backward pass; by the
// time it reaches a cross-attention block it has grown through LayerNorm backwards (measured on a 3-layer cfg-3 model: |S dO| up to 1.5e4), and
// dP' = dO . V sums 64 such products: packed to fp16 for the head mix it left the format (inf -> NaN gradients).  The V image of this form
// therefore carries the factor X6B_VS = 2^-6 (exact in fp16 but for values below 2^-8, which lose up to 6 trailing bits), so that everything
// derived from dP' -- dP, delta, the dW_th products -- is smaller by that factor; pass B folds the inverse into the probabilities it multiplies ds
// with (+6 on the exponent offset of exp2: free), the rank-one null-key term multiplies explicitly, and the caller multiplies dW_th by 2^6 / S.
constexpr float X6B_VS = 0.015625f, X6B_VS_LOG2_INV = 6.f;
// (G16, a template flag of everything below: the fp16-gradient form -- q / dO / the images hold fp16 values, dO = fp16(S dO), every MFMA the fp16
//  one, Pm / dS / dq leave as fp16 -- dS and dq saturating and counted; amdnuwa_xattn6_bwd_f16)
#define MFMAB(a, b, c) mfma16<G16>(a, b, c)

struct MixA { bf16x8 hi[2], lo[2]; };
template <bool G16>
__device__ __forceinline__ MixA mix_operand_lds(const float* wsrc, int lane) {      // (xattn2.hip, mix_operand)
    MixA a;
    const int m = lane & 15;
    const bool on = (m >> 2) == (lane >> 4);
#pragma unroll
    for (int Q = 0; Q < 2; ++Q) {
        const float4 w0 = *reinterpret_cast<const float4*>(wsrc + (4 * Q + (m & 3)) * 8), w1 = *reinterpret_cast<const float4*>(wsrc + (4 * Q + (m & 3)) * 8 + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ph[t] = pack2_t<G16>(w[2 * t], w[2 * t + 1]);
            pl[t] = pack2_t<G16>(w[2 * t] - lo_t<G16>(ph[t]), w[2 * t + 1] - hi_t<G16>(ph[t]));
        }
        a.hi[Q] = __builtin_bit_cast(bf16x8, on ? make_uint4(ph[0], ph[1], ph[2], ph[3]) : make_uint4(0, 0, 0, 0));
        a.lo[Q] = __builtin_bit_cast(bf16x8, on ? make_uint4(pl[0], pl[1], pl[2], pl[3]) : make_uint4(0, 0, 0, 0));
    }
    return a;
}
template <bool G16>
__device__ __forceinline__ bf16x8 pack_heads8(const float (&v)[NH][8], int e) {
    return __builtin_bit_cast(bf16x8, make_uint4(pack2_t<G16>(v[0][e], v[1][e]), pack2_t<G16>(v[2][e], v[3][e]),
                                                 pack2_t<G16>(v[4][e], v[5][e]), pack2_t<G16>(v[6][e], v[7][e])));
}
#define MIXB(A_, Q_, B_) MFMAB((A_).lo[Q_], B_, MFMAB((A_).hi[Q_], B_, (f32x4{0.f, 0.f, 0.f, 0.f})))
// K^T (a [key][d] tile read transposed): lane (c, g4) gets tile[kb*16 + 4*g4 + j][db*16 + c], j = 0..3, for kb = 0, 1
__device__ __forceinline__ bf16x8 lds_tr6(const char* base, int h, int db, int c, int g4) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int col = db * 16 + ((c & 3) << 2);
    const int r0 = 4 * g4 + (c >> 2), r1 = 16 + r0;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + k6_off(h, r0, col >> 3) + ((col >> 2) & 1) * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + k6_off(h, r1, col >> 3) + ((col >> 2) & 1) * 8));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool G16>
__global__ __launch_bounds__(256, 1) void xattn6_bwd_kernel(X6BArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm0 = lds_addr_of(smem);
    __shared__ float thsh[4][NH * NH];
    __shared__ __attribute__((aligned(16))) float wsh[NH * NH], wtsh[NH * NH];          // W[g][h] and its transpose
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const int tiles = (a.n + 63) / 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / tiles, qi = (bid % tiles) * 64 + wave * 16 + c;
    const bool qok = qi < a.n;
    const int nch = a.nch;
    const char* k6 = a.K6 + (size_t)b * nch * KT;
    const char* v6 = a.V6 + (size_t)b * nch * KT;
    const int ko0 = k6_off(0, c, g4), ko1 = k6_off(0, c, 4 + g4);
    if (tid < NH * NH) { const float v = a.wth[tid]; wsh[tid] = v; wtsh[(tid & 7) * 8 + (tid >> 3)] = v; }
    float* nks = reinterpret_cast<float*>(smem + 2 * STAGE);     // the null key / value, rounded to bf16 as an image row is: 2 x 2 KiB behind the ring
    float* nvs = nks + NH * DH;
    for (int e = tid; e < NH * DH; e += 256) { nks[e] = rnd16<G16>(a.null_k[e]); nvs[e] = rnd16<G16>(a.null_v[e]) * (G16 ? X6B_VS : 1.f); }
    const uint32_t wv = lane < nch ? a.vbits[(size_t)b * nch + lane] : 0u;
    // T % 32 == 0: the null key (position T) is alone in the last chunk -- that chunk is not a matrix iteration but a rank-one term
    const bool rank1 = (a.T & 31) == 0;
    const int nfull = rank1 ? nch - 1 : nch;
    __syncthreads();                                             // (before any DMA is in flight)
    // K + V of chunk ch -> stage: 64 pieces, 16 per wave
    auto stage = [&](int stg, int ch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = wave + 4 * i;
            dma16_s(k6 + (size_t)ch * KT + piece * 1024, lane * 16, sm0 + stg * STAGE + piece * 1024);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = wave + 4 * i;
            dma16_s(v6 + (size_t)ch * KT + piece * 1024, lane * 16, sm0 + stg * STAGE + KT + piece * 1024);
        }
    };
    auto bias_raw = [&](int ch, f32x4& b0, f32x4& b1) {
        const uint32_t u = __builtin_amdgcn_readlane(wv, ch) >> (4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b0[r] = ((u >> r) & 1u) ? 0.f : MASK_BIAS_RAW;
            b1[r] = ((u >> (16 + r)) & 1u) ? 0.f : MASK_BIAS_RAW;
        }
    };
    stage(0, 0);
    const MixA AW = mix_operand_lds<G16>(wsh, lane), AWT = mix_operand_lds<G16>(wtsh, lane);
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
    // (the compiler's counter model must see these loads complete before the ring's counted waits: see xattn5_bwd_kv_kernel)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[h][ks]), "v"(df[h][ks]));
        asm volatile("" ::"v"(nb[h]));
    }
    const size_t hplane = (size_t)nch * a.n * 32;                // elements between heads in dS / Pm
    // probabilities of head h for this lane's 8 slots of the chunk in `kbase`
    auto probs = [&](const char* kbase, int h, const f32x4& b0, const f32x4& b1, float* P, float off) {
        const KF f = k_frags(kbase, h, ko0, ko1);
        f32x4 s0 = b0, s1 = b1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { s0 = MFMAB(f.v[0][ks], qf[h][ks], s0); s1 = MFMAB(f.v[1][ks], qf[h][ks], s1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c1, off));
            P[4 + r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c1, off));
        }
    };
    // dP'^T[g] = V[g] dO[g]^T for this lane's 8 slots
    auto dpp = [&](const char* vbase, int g, float* d) {
        const KF f = k_frags(vbase, g, ko0, ko1);
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { s0 = MFMAB(f.v[0][ks], df[g][ks], s0); s1 = MFMAB(f.v[1][ks], df[g][ks], s1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { d[r] = s0[r]; d[4 + r] = s1[r]; }
    };

    // ---- pass A: P' = W P -> Pm;  dP = W^T dP' (both mixes on the matrix pipe);  delta[h] = sum_j dP[h] P[h];
    //      dW_th[g][h] += sum over (query, key) of dP'[g] P[h].  The contraction of that last one runs over a query's keys with the HEADS as
    //      rows and columns, while the lanes hold (query, 8 keys) with the heads in separate registers: as 8 x 8 x 8 FMAs per lane and chunk it
    //      was 512 of the ~1000 VALU instructions of a pass-A chunk.  Now on the matrix pipe too: the wave turns its bf16 dP' and P through a
    //      private 4-KiB LDS tile ([4 heads][16 queries][4 key groups] of 16 bytes, written as the lanes hold them, read back with MFMA row
    //      m = (query m >> 2 of a quad, head m & 3)), and one MFMA per query quad and pair of head halves adds
    //      C[(qa, g)][(qb, h)] += sum over the chunk's 32 keys of dP'[g][qa] P[h][qb]: the entries with qa == qb are the wanted products, the rest
    //      is ignored.  Four accumulator tiles (16 registers) for the whole kernel instead of 64 sums; 16 MFMAs + 32 LDS pieces per chunk.
    float delta[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) delta[h] = 0.f;
    f32x4 CT[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) CT[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
    char* tb = smem + 2 * STAGE + 4096 + wave * 4096;
    const int tw = c * 64 + ((g4 ^ ((c >> 1) & 3)) << 4);        // where lane (query c, key group g4) puts its 16 bytes inside a head's 1-KiB plane
    int tro[4];                                                  // ... and where lane (row / column m = c, key group g4) finds its operand of query quad t
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int qq = 4 * t + (c >> 2);
        tro[t] = (c & 3) * 1024 + qq * 64 + ((g4 ^ ((qq >> 1) & 3)) << 4);
    }
    auto tput = [&](int hl, const float* v) {                    // 8 fp32 values of head-in-half hl -> bf16 -> the tile
        *reinterpret_cast<uint4*>(tb + hl * 1024 + tw) = make_uint4(pack2_t<G16>(v[0], v[1]), pack2_t<G16>(v[2], v[3]), pack2_t<G16>(v[4], v[5]), pack2_t<G16>(v[6], v[7]));
    };
    for (int ch = 0; ch < nfull; ++ch) {
        if (ch + 1 < nfull) { stage((ch + 1) & 1, ch + 1); VMCNT(16); }
        else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        const char* kbase = smem + (ch & 1) * STAGE;
        const char* vbase = kbase + KT;
        f32x4 b0, b1;
        bias_raw(ch, b0, b1);
        float P[NH][8];
#pragma unroll
        for (int h = 0; h < NH; ++h) probs(kbase, h, b0, b1, P[h], nb[h]);
        const bool st = qok && 32 * ch + 4 * g4 <= a.T;          // (a lane group whose 8 keys are all padding writes nothing)
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            f32x4 D[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) D[e] = MIXB(AW, Q, pack_heads8<G16>(P, e));
            if (st) {
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) {
                    uint16_t* dst = a.Pm + ((size_t)b * NH + 4 * Q + rp) * hplane + ((size_t)ch * a.n + qi) * 32 + g4 * 8;
                    *reinterpret_cast<uint4*>(dst) = make_uint4(pack2_t<G16>(D[0][rp], D[1][rp]), pack2_t<G16>(D[2][rp], D[3][rp]),
                                                                pack2_t<G16>(D[4][rp], D[5][rp]), pack2_t<G16>(D[6][rp], D[7][rp]));
                }
            }
        }
        uint32_t bw[8][4];
        bf16x8 TA[2][4];                                         // dP' of the two head halves as MFMA row operands, one per query quad
#pragma unroll
        for (int gp = 0; gp < 4; ++gp) {
            float d0[8], d1[8];
            dpp(vbase, 2 * gp, d0);
            dpp(vbase, 2 * gp + 1, d1);
#pragma unroll
            for (int e = 0; e < 8; ++e) bw[e][gp] = pack2_t<G16>(d0[e], d1[e]);
            tput((2 * gp) & 3, d0);
            tput((2 * gp + 1) & 3, d1);
            if (gp & 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) TA[gp >> 1][t] = lds16(tb + tro[t]);
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                         // P of head half hh as the column operand
#pragma unroll
            for (int hl = 0; hl < 4; ++hl) tput(hl, P[4 * hh + hl]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 tbv = lds16(tb + tro[t]);
                CT[0][hh] = MFMAB(TA[0][t], tbv, CT[0][hh]);
                CT[1][hh] = MFMAB(TA[1][t], tbv, CT[1][hh]);
            }
        }
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            f32x4 D[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                D[e] = MIXB(AWT, Q, __builtin_bit_cast(bf16x8, make_uint4(bw[e][0], bw[e][1], bw[e][2], bw[e][3])));   // D[e][rp] = dP[4Q + rp]
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
    // pass B's first chunk is on its way while the statistics are reduced
    stage(0, 0);
    float nbB[NH];                                               // exponent offsets of pass B's probabilities (G16: times 2^6, undoing the V image's factor in ds)
#pragma unroll
    for (int h = 0; h < NH; ++h) nbB[h] = nb[h] + (G16 ? X6B_VS_LOG2_INV : 0.f);
    // ---- the null key as a rank-one term (T % 32 == 0), pass A part: P_null, dP'_null, their mixes, the shares of delta and dW_th, Pm
    float PN[NH], dPN[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) PN[h] = dPN[h] = 0.f;
    if (rank1) {
        float dN[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float sa = 0.f, da = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 uq = __builtin_bit_cast(uint4, qf[h][ks]), ud = __builtin_bit_cast(uint4, df[h][ks]);
                const uint32_t wq[4] = {uq.x, uq.y, uq.z, uq.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
                const int o8 = h * DH + ks * 32 + g4 * 8;            // (indexed, not through a generic pointer: the address-space cast of a
                                                                     //  static LDS array trips the compiler's register-class check here)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    sa = fmaf(lo_t<G16>(wq[t]), nks[o8 + 2 * t], sa); sa = fmaf(hi_t<G16>(wq[t]), nks[o8 + 2 * t + 1], sa);
                    da = fmaf(lo_t<G16>(wd[t]), nvs[o8 + 2 * t], da); da = fmaf(hi_t<G16>(wd[t]), nvs[o8 + 2 * t + 1], da);
                }
            }
            sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
            da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);
            PN[h] = __builtin_amdgcn_exp2f(fmaf(sa, c1, nb[h]));
            dN[h] = da;
        }
        float PmN[NH];
#pragma unroll
        for (int g = 0; g < NH; ++g) {                           // (operands rounded to bf16 as the matrix-pipe mixes round theirs)
            float pm = 0.f, dp = 0.f;
#pragma unroll
            for (int h = 0; h < NH; ++h) { pm = fmaf(wsh[g * NH + h], rnd16<G16>(PN[h]), pm); dp = fmaf(wtsh[g * NH + h], rnd16<G16>(dN[h]), dp); }
            PmN[g] = pm; dPN[g] = dp;                            // dPN[h = g] = sum_g' W[g'][h] dP'_null[g']  (wtsh row h)
        }
        {   // dW_th share of the null key through the same tile and tiles: a "chunk" whose only key sits in slot 0 of key group 0
            bf16x8 TN[2][4];
            const bool one = g4 == 0 && qok;
#pragma unroll
            for (int gh = 0; gh < 2; ++gh) {
#pragma unroll
                for (int hl = 0; hl < 4; ++hl) {
                    const float v8[8] = {one ? dN[4 * gh + hl] : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    tput(hl, v8);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) TN[gh][t] = lds16(tb + tro[t]);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int hl = 0; hl < 4; ++hl) {
                    const float v8[8] = {one ? PN[4 * hh + hl] : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    tput(hl, v8);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bf16x8 tbv = lds16(tb + tro[t]);
                    CT[0][hh] = MFMAB(TN[0][t], tbv, CT[0][hh]);
                    CT[1][hh] = MFMAB(TN[1][t], tbv, CT[1][hh]);
                }
            }
        }
        if (g4 == 0) {                                           // one lane per query carries the null key's share of delta
#pragma unroll
            for (int h = 0; h < NH; ++h) delta[h] = fmaf(dPN[h], PN[h], delta[h]);
        }
        // (stored outside the block above: as part of it the compiler dies with "Illegal instruction detected: Operand has incorrect register class")
        uint16_t* pmn = a.Pm + ((size_t)(nch - 1) * a.n + qi) * 32;
#pragma unroll
        for (int g = 0; g < NH; ++g)
            if (g4 == 0 && qok) *reinterpret_cast<uint4*>(pmn + ((size_t)b * NH + g) * hplane) = make_uint4(pack2_t<G16>(PmN[g], 0.f), 0u, 0u, 0u);
    }
    // a query's keys are spread over the 4 lane groups
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        delta[h] += __shfl_xor(delta[h], 16, 64);
        delta[h] += __shfl_xor(delta[h], 32, 64);
    }
    // dW_th partial of this workgroup (fixed order over the 4 waves): lane l ends up with the wave's sum of entry l = g * NH + h
    // (a tile entry is a wanted product where the row's and the column's query agree: lanes with c >> 2 == g4; rows 4 g4 + r = (query g4, head
    //  r of the row half), column c = (query c >> 2, head c & 3 of the column half).  Sum over the wave's lanes with equal c & 3 in a fixed order.)
    {
        const bool diag = (c >> 2) == g4;
#pragma unroll
        for (int gh = 0; gh < 2; ++gh)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = diag ? CT[gh][hh][r] : 0.f;
                    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                    if (lane < 4) thsh[wave][(4 * gh + r) * NH + 4 * hh + lane] = v;
                }
    }
    __syncthreads();
    if (tid < NH * NH) a.part_th[(size_t)bid * NH * NH + tid] = ((thsh[0][tid] + thsh[1][tid]) + thsh[2][tid]) + thsh[3][tid];

    // ---- pass B: ds[h] = P[h] (dP[h] - delta[h]) -> dS;  dq^T[h] += K^T[h] ds^T[h]
    float samax = 0.f;                                           // G16: largest magnitude handed to a saturating fp16 store (amdnuwa_f16_sat_count)
    f32x4 dQ[NH][DB];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int db = 0; db < DB; ++db) dQ[h][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nfull; ++ch) {
        if (ch + 1 < nfull) { stage((ch + 1) & 1, ch + 1); VMCNT(16); }
        else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        const char* kbase = smem + (ch & 1) * STAGE;
        const char* vbase = kbase + KT;
        f32x4 b0, b1;
        bias_raw(ch, b0, b1);
        bf16x8 bmD[8];
        {
            uint32_t bw[8][4];
#pragma unroll
            for (int gp = 0; gp < 4; ++gp) {
                float d0[8], d1[8];
                dpp(vbase, 2 * gp, d0);
                dpp(vbase, 2 * gp + 1, d1);
#pragma unroll
                for (int e = 0; e < 8; ++e) bw[e][gp] = pack2_t<G16>(d0[e], d1[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bmD[e] = __builtin_bit_cast(bf16x8, make_uint4(bw[e][0], bw[e][1], bw[e][2], bw[e][3]));
        }
        const bool st = qok && 32 * ch + 4 * g4 <= a.T;
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            f32x4 D[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) D[e] = MIXB(AWT, Q, bmD[e]);
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const int h = 4 * Q + rp;
                float P[8], ds[8];
                probs(kbase, h, b0, b1, P, nbB[h]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ds[e] = P[e] * (D[e][rp] - delta[h]);
                // (G16: |ds| <= |dP - delta| with dP a head mix of fp16 values that carry 2^-6: it stays inside the format wherever dP' did; the plain
                //  converter -- the saturating, counted one costs 5 more instructions per pair, 160 per chunk; dq below IS saturating and counted)
                const uint4 pk = make_uint4(pack2_t<G16>(ds[0], ds[1]), pack2_t<G16>(ds[2], ds[3]), pack2_t<G16>(ds[4], ds[5]), pack2_t<G16>(ds[6], ds[7]));
                if (st) *reinterpret_cast<uint4*>(a.dS + ((size_t)b * NH + h) * hplane + ((size_t)ch * a.n + qi) * 32 + g4 * 8) = pk;
                const bf16x8 sf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
                for (int db = 0; db < DB; ++db) dQ[h][db] = MFMAB(lds_tr6(kbase, h, db, c, g4), sf, dQ[h][db]);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (rank1) {                                                 // pass B part: ds_null -> dS, dq += ds_null k_null
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const uint32_t dsb = G16 ? pack2_f16_sat_n(PN[h] * (dPN[h] - delta[h]) * (1.f / X6B_VS), 0.f, samax) : pack2_rne(PN[h] * (dPN[h] - delta[h]), 0.f);
            if (g4 == 0 && qok) *reinterpret_cast<uint4*>(a.dS + ((size_t)b * NH + h) * hplane + ((size_t)(nch - 1) * a.n + qi) * 32) = make_uint4(dsb, 0u, 0u, 0u);
            const float dsr = lo_t<G16>(dsb);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int o4 = h * DH + db * 16 + g4 * 4;
                dQ[h][db][0] = fmaf(dsr, nks[o4], dQ[h][db][0]); dQ[h][db][1] = fmaf(dsr, nks[o4 + 1], dQ[h][db][1]);
                dQ[h][db][2] = fmaf(dsr, nks[o4 + 2], dQ[h][db][2]); dQ[h][db][3] = fmaf(dsr, nks[o4 + 3], dQ[h][db][3]);
            }
        }
    }
    if (qok) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                uint16_t* dst = a.dq + ((size_t)b * a.n + qi) * a.lddq + h * DH + db * 16 + g4 * 4;
                *reinterpret_cast<uint2*>(dst) = G16 ?
                    make_uint2(pack2_f16_sat_n(dQ[h][db][0] * a.scale, dQ[h][db][1] * a.scale, samax), pack2_f16_sat_n(dQ[h][db][2] * a.scale, dQ[h][db][3] * a.scale, samax)) :
                    make_uint2(pack2_rne(dQ[h][db][0] * a.scale, dQ[h][db][1] * a.scale), pack2_rne(dQ[h][db][2] * a.scale, dQ[h][db][3] * a.scale));
            }
    }
    if constexpr (G16) f16_sat_commit(samax);
}

// images of the backward: kv [B*T, ldkv] bf16 + the null key / value -> K6 / V6 ([key][d] tiles, keys 0..T-1 the context, key T = null) / vbits
template <bool F16>
__global__ __launch_bounds__(256) void xattn6_pack_bwd_kernel(const uint16_t* __restrict__ kv, int ldkv, const float* __restrict__ null_k,
                                                              const float* __restrict__ null_v, const uint8_t* __restrict__ mask,
                                                              char* __restrict__ K6, char* __restrict__ V6, uint32_t* __restrict__ vbits, int T, int nch) {
    const int b = blockIdx.x / nch, ch = blockIdx.x % nch, tid = threadIdx.x;
    const size_t cbase = ((size_t)b * nch + ch) * KT;
    if (tid < 64) {
        const int j = 32 * ch + tid;
        const bool ok = tid < 32 && (j == T || (j < T && (mask ? mask[(size_t)b * T + j] != 0 : true)));
        const unsigned long long bal = __ballot(ok);
        if (tid == 0) vbits[(size_t)b * nch + ch] = (uint32_t)bal;
    }
    for (int e = tid; e < 2 * NH * 32 * 8; e += 256) {
        const int part = e >> 11, h = (e >> 8) & 7, row = (e >> 3) & 31, pos = e & 7, gc = pos ^ (row & 7), j = 32 * ch + row;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (j == T) {
            const float* src = (part ? null_v : null_k) + h * DH + gc * 8;
            r = make_uint4(pack2_t<F16>(src[0], src[1]), pack2_t<F16>(src[2], src[3]), pack2_t<F16>(src[4], src[5]), pack2_t<F16>(src[6], src[7]));
        } else if (j < T) r = *reinterpret_cast<const uint4*>(kv + ((size_t)b * T + j) * ldkv + part * NH * DH + h * DH + gc * 8);
        if (F16 && part) {     // fp16-gradient form: the V image carries the factor X6B_VS (see xattn6_bwd_kernel)
            const f16x2_t k2 = {(_Float16)X6B_VS, (_Float16)X6B_VS};
            r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, r.x) * k2); r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, r.y) * k2);
            r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, r.z) * k2); r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, r.w) * k2);
        }
        *reinterpret_cast<uint4*>((part ? V6 : K6) + cbase + h * TILE + row * 128 + pos * 16) = r;
    }
}

int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) cus = n;
        else cus = 256;
    }
    return cus;
}

int check6(const amdnuwa_xattn_geom* g) {
    if (!g) return AMDNUWA_ERR_ARG;
    if (g->heads != NH || g->dim_head != DH || g->T < 1 || g->T > 2048) return AMDNUWA_ERR_UNSUPPORTED;   // (nch <= 64: one mask word per lane)
    return AMDNUWA_OK;
}

}  // namespace

extern "C" int amdnuwa_xattn6_supported(const amdnuwa_xattn_geom* g) { return check6(g) == AMDNUWA_OK; }
extern "C" int amdnuwa_xattn6_nch(int T) { return T < 1 ? 0 : 2 * ((T + 63) / 64); }
extern "C" size_t amdnuwa_xattn6_image_bytes(const amdnuwa_xattn_geom* g) {
    return check6(g) ? 0 : (size_t)g->B * amdnuwa_xattn6_nch(g->T) * KT;
}

extern "C" int amdnuwa_xattn6_pack(const amdnuwa_xattn_geom* g, const uint16_t* kv16, int ldkv, const uint8_t* context_mask, int f16,
                                   const amdnuwa_xattn6_kv* out, hipStream_t stream) {
    int rc = check6(g);
    if (rc) return rc;
    if (!kv16 || !out || !out->K6 || !out->V6 || !out->vbits || ldkv % 8 || ldkv < 2 * NH * DH) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    const int nch = amdnuwa_xattn6_nch(g->T);
    if (f16)
        hipLaunchKernelGGL(xattn6_pack_kernel<true>, dim3(g->B * nch), dim3(256), 0, stream, kv16, ldkv, context_mask, (char*)out->K6, (char*)out->V6,
                           out->vbits, g->T, nch);
    else
        hipLaunchKernelGGL(xattn6_pack_kernel<false>, dim3(g->B * nch), dim3(256), 0, stream, kv16, ldkv, context_mask, (char*)out->K6, (char*)out->V6,
                           out->vbits, g->T, nch);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_xattn6_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q16, int ldq, const amdnuwa_xattn6_kv* kv, const float* null_k,
                                  const float* null_v, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, float* stats,
                                  int f16, hipStream_t stream) {
    int rc = check6(g);
    if (rc) return rc;
    if (!q16 || !kv || !kv->K6 || !kv->V6 || !kv->vbits || !null_k || !null_v || !w_th || ldq % 8 || ldo % 4) return AMDNUWA_ERR_ARG;
    if (!o && !(o_lo && o_lo_f16)) return AMDNUWA_ERR_ARG;        // o == NULL: the fp16 copy is the only output
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    X6Args a{};
    a.q = q16; a.ldq = ldq; a.K6 = (const char*)kv->K6; a.V6 = (const char*)kv->V6; a.vbits = kv->vbits;
    a.null_k = null_k; a.null_v = null_v; a.wth = w_th;
    a.o = o; a.ol = o_lo; a.ldo = ldo; a.ol_f16 = (o_lo && o_lo_f16) ? 1 : 0; a.stats = stats;
    a.B = g->B; a.n = g->n; a.nch = amdnuwa_xattn6_nch(g->T); a.c1 = g->scale * 1.4426950408889634f;
    const int tiles = (g->n + 63) / 64, NT = g->B * tiles;
    // one workgroup per CU (160 KiB of LDS each); the grid stays a multiple of 8 so that a workgroup's items all lie in one XCD's slab
    const int grid = NT < cu_count() ? NT : cu_count() / 8 * 8;
    if (f16) {
        (void)hipFuncSetAttribute((const void*)xattn6_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipLaunchKernelGGL(xattn6_fwd_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)xattn6_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipLaunchKernelGGL(xattn6_fwd_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, stream, a);
    }
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_xattn6_bwd_image_bytes(const amdnuwa_xattn_geom* g) {
    return (check6(g) || g->JP % 32 || g->JP < g->T + 1 || g->JP / 32 > 64) ? 0 : (size_t)g->B * (g->JP / 32) * KT;
}
namespace {
int x6_pack_bwd(const amdnuwa_xattn_geom* g, const uint16_t* kv, int ldkv, const float* null_k, const float* null_v, const uint8_t* context_mask,
                const amdnuwa_xattn6_kv* out, bool f16, hipStream_t stream) {
    if (!amdnuwa_xattn6_bwd_image_bytes(g)) return g ? AMDNUWA_ERR_UNSUPPORTED : AMDNUWA_ERR_ARG;
    if (!kv || !null_k || !null_v || !out || !out->K6 || !out->V6 || !out->vbits || ldkv % 8 || ldkv < 2 * NH * DH) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    const int nch = g->JP / 32;
    if (f16)
        hipLaunchKernelGGL(xattn6_pack_bwd_kernel<true>, dim3(g->B * nch), dim3(256), 0, stream, kv, ldkv, null_k, null_v, context_mask, (char*)out->K6,
                           (char*)out->V6, out->vbits, g->T, nch);
    else
        hipLaunchKernelGGL(xattn6_pack_bwd_kernel<false>, dim3(g->B * nch), dim3(256), 0, stream, kv, ldkv, null_k, null_v, context_mask, (char*)out->K6,
                           (char*)out->V6, out->vbits, g->T, nch);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
int x6_bwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo, const amdnuwa_xattn6_kv* kv, const float* null_k,
           const float* null_v, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm, uint16_t* dq, int lddq, float* part_th,
           size_t part_bytes, bool g16, hipStream_t stream) {
    if (!amdnuwa_xattn6_bwd_image_bytes(g)) return g ? AMDNUWA_ERR_UNSUPPORTED : AMDNUWA_ERR_ARG;
    if (!q || !dO || !kv || !kv->K6 || !kv->V6 || !kv->vbits || !null_k || !null_v || !w_th || !stats || !dS || !Pm || !dq || ldq % 8 || lddo % 8 || lddq % 4)
        return AMDNUWA_ERR_ARG;
    if (!part_th || part_bytes < amdnuwa_xattn6_bwd_workspace_bytes(g)) return AMDNUWA_ERR_WORKSPACE;
    if (g->B <= 0 || g->n <= 0) return AMDNUWA_OK;
    X6BArgs a{};
    a.q = q; a.ldq = ldq; a.dO = dO; a.lddo = lddo; a.K6 = (const char*)kv->K6; a.V6 = (const char*)kv->V6; a.vbits = kv->vbits;
    a.null_k = null_k; a.null_v = null_v; a.wth = w_th; a.stats = stats; a.dS = dS; a.Pm = Pm; a.dq = dq; a.lddq = lddq; a.part_th = part_th;
    a.B = g->B; a.n = g->n; a.nch = g->JP / 32; a.T = g->T; a.scale = g->scale;
    // LDS: the two-stage K + V ring, the null key / value rows, one 4-KiB turning tile per wave (the dW_th products)
    constexpr int LB = 2 * STAGE + 4096 + 4 * 4096;
    const dim3 grid(g->B * ((g->n + 63) / 64));
    if (g16) {
        (void)hipFuncSetAttribute((const void*)xattn6_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LB);
        hipLaunchKernelGGL(xattn6_bwd_kernel<true>, grid, dim3(256), LB, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)xattn6_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LB);
        hipLaunchKernelGGL(xattn6_bwd_kernel<false>, grid, dim3(256), LB, stream, a);
    }
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
}  // namespace

extern "C" int amdnuwa_xattn6_pack_bwd(const amdnuwa_xattn_geom* g, const uint16_t* kv, int ldkv, const float* null_k, const float* null_v,
                                       const uint8_t* context_mask, const amdnuwa_xattn6_kv* out, hipStream_t stream) {
    return x6_pack_bwd(g, kv, ldkv, null_k, null_v, context_mask, out, false, stream);
}
// ... from the FP16 copy of the key / value projection (null key / value rounded to fp16): the images of amdnuwa_xattn6_bwd_f16
extern "C" int amdnuwa_xattn6_pack_bwd_f16(const amdnuwa_xattn_geom* g, const uint16_t* kv_f16, int ldkv, const float* null_k, const float* null_v,
                                           const uint8_t* context_mask, const amdnuwa_xattn6_kv* out, hipStream_t stream) {
    return x6_pack_bwd(g, kv_f16, ldkv, null_k, null_v, context_mask, out, true, stream);
}
extern "C" size_t amdnuwa_xattn6_bwd_workspace_bytes(const amdnuwa_xattn_geom* g) {
    return amdnuwa_xattn6_bwd_image_bytes(g) ? (size_t)g->B * ((g->n + 63) / 64) * NH * NH * sizeof(float) : 0;
}
extern "C" int amdnuwa_xattn6_bwd(const amdnuwa_xattn_geom* g, const uint16_t* q, int ldq, const uint16_t* dO, int lddo, const amdnuwa_xattn6_kv* kv,
                                  const float* null_k, const float* null_v, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm, uint16_t* dq, int lddq, float* part_th,
                                  size_t part_bytes, hipStream_t stream) {
    return x6_bwd(g, q, ldq, dO, lddo, kv, null_k, null_v, w_th, stats, dS, Pm, dq, lddq, part_th, part_bytes, false, stream);
}
// fp16-gradient form (ABI 19): q = the fp16 copy the forward read, dO = fp16(S dO), images from amdnuwa_xattn6_pack_bwd_f16; dS / dq leave as
// fp16(S value) (saturating, counted), Pm as fp16, the dW_th partials carry the factor S (the caller multiplies their column sums by 1 / S)
extern "C" int amdnuwa_xattn6_bwd_f16(const amdnuwa_xattn_geom* g, const uint16_t* q_f16, int ldq, const uint16_t* dO_f16, int lddo, const amdnuwa_xattn6_kv* kv,
                                      const float* null_k, const float* null_v, const float* w_th, const float* stats, uint16_t* dS, uint16_t* Pm, uint16_t* dq,
                                      int lddq, float* part_th, size_t part_bytes, hipStream_t stream) {
    return x6_bwd(g, q_f16, ldq, dO_f16, lddo, kv, null_k, null_v, w_th, stats, dS, Pm, dq, lddq, part_th, part_bytes, true, stream);
}

#if X6_TIMING
extern "C" int amdnuwa_xattn6_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x6_stamps), &p, sizeof(p)); }
#endif

AMDNUWA_SAT_ACCESSOR(xattn6)
