// Text cross-attention core, third design ("xattn6"): np.py:339-378 for 8 heads x dim_head 64, any context length.
//
// What changed against xattn4 (xattn2.hip), and why (round 6; the ISA of xattn4 showed one exposed LDS round trip per head in the
// score sweep, one per slot in the head mix, and 6 register moves per P'V operand -- the kernel ran at 6 % of the MFMA peak although
// neither the matrix pipe, the VALU nor the LDS was busy):
//   * K / V travel as IMAGES IN LDS ORDER written once per layer by amdnuwa_xattn6_pack: a 32-key chunk of all heads is 32 KiB of K
//     ([head][key][d], bank swizzle baked in) and 32 KiB of V^T ([head][d][key slot], key slots in the order a lane holds its 8
//     probabilities, swizzle baked in): staging is a linear copy (1 KiB DMA pieces, no per-lane address arithmetic) and EVERY MFMA
//     operand is ONE conflict-free ds_read_b128;
//   * K is stored pre-multiplied by scale * log2(e): the scores come out of the MFMA in the log2 domain;
//   * the key mask is the C operand of the first score MFMA (0 or MASK_BIAS per key row, built from one 32-bit word per chunk that
//     arrives through the scalar cache): masking costs no per-element VALU work;
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
constexpr float MASK_BIAS = -60000.f;        // log2-domain score of a masked key: exp2(MASK_BIAS - max) == 0

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
__device__ __forceinline__ void dma16_s(const char* sbase, uint32_t voff, void* lds) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr_t)lds);
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
            dma16_s(img + (size_t)ch * KT + piece * 1024, lane * 16, smem + slot * KT + piece * 1024);
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
                const float4 k0 = *reinterpret_cast<const float4*>(a.null_k + (H0 + h) * DH + ks * 32 + g4 * 8);
                const float4 k1 = *reinterpret_cast<const float4*>(a.null_k + (H0 + h) * DH + ks * 32 + g4 * 8 + 4);
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
                const float mn = fmaxf(m[h], fmaxf(ca, cb));
                float acc = l[h] * __builtin_amdgcn_exp2f(m[h] - mn);
                float acc2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc += __builtin_amdgcn_exp2f(s0v[r] - mn);
                    acc2 += __builtin_amdgcn_exp2f(s1v[r] - mn);
                    acc += __builtin_amdgcn_exp2f(t0[r] - mn);
                    acc2 += __builtin_amdgcn_exp2f(t1[r] - mn);
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
                    for (int h = 0; h < NHH; ++h) pe[h] = __builtin_amdgcn_exp2f((e < 4 ? s0v[h][e & 3] : s1v[h][e & 3]) + nb[h]);
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
#pragma unroll
            for (int rp = 0; rp < NHH; ++rp)
#pragma unroll
                for (int dp = 0; dp < DB; dp += 2) {
                    const size_t go = ((size_t)b * a.n + qi) * a.ldo + (H0 + rp) * DH + dp * 16 + dlane;
                    uint32_t x0 = pack2_rne(O[rp][dp][0], O[rp][dp][1]), x1 = pack2_rne(O[rp][dp][2], O[rp][dp][3]);
                    uint32_t y0 = pack2_rne(O[rp][dp + 1][0], O[rp][dp + 1][1]), y1 = pack2_rne(O[rp][dp + 1][2], O[rp][dp + 1][3]);
                    uint32_t lx0, lx1, ly0, ly1;                  // the second output: fp16 copy or bf16 residual
                    if (a.ol_f16) {
                        lx0 = pack2_f16_sat(O[rp][dp][0], O[rp][dp][1]); lx1 = pack2_f16_sat(O[rp][dp][2], O[rp][dp][3]);
                        ly0 = pack2_f16_sat(O[rp][dp + 1][0], O[rp][dp + 1][1]); ly1 = pack2_f16_sat(O[rp][dp + 1][2], O[rp][dp + 1][3]);
                    } else {
                        lx0 = pack2_rne(O[rp][dp][0] - lo_f(x0), O[rp][dp][1] - hi_f(x0)); lx1 = pack2_rne(O[rp][dp][2] - lo_f(x1), O[rp][dp][3] - hi_f(x1));
                        ly0 = pack2_rne(O[rp][dp + 1][0] - lo_f(y0), O[rp][dp + 1][1] - hi_f(y0)); ly1 = pack2_rne(O[rp][dp + 1][2] - lo_f(y1), O[rp][dp + 1][3] - hi_f(y1));
                    }
                    auto swap = [](uint32_t& x, uint32_t& y) {
                        const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
                        x = r[0]; y = r[1];
                    };
                    swap(x0, y0); swap(x1, y1);
                    if (qok) *reinterpret_cast<uint4*>(a.o + go) = make_uint4(x0, x1, y0, y1);
                    if (a.ol) {
                        swap(lx0, ly0); swap(lx1, ly1);
                        if (qok) *reinterpret_cast<uint4*>(a.ol + go) = make_uint4(lx0, lx1, ly0, ly1);
                    }
                }
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
                                                          int T, int nch, float c1) {
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
    // K image: piece (h, row, pos) holds the 8 d values of chunk gc = pos ^ (row & 7), times c1
    for (int e = tid; e < NH * 32 * 8; e += 256) {
        const int h = e >> 8, row = (e >> 3) & 31, pos = e & 7, gc = pos ^ (row & 7), j = 32 * ch + row;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (j < T) {
            const uint4 u = *reinterpret_cast<const uint4*>(kv + ((size_t)b * T + j) * ldkv + h * DH + gc * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            uint32_t o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float x0 = h2f<F16>((uint16_t)(w[t] & 0xffff)) * c1, x1 = h2f<F16>((uint16_t)(w[t] >> 16)) * c1;
                o[t] = F16 ? pack2_f16_sat(x0, x1) : pack2_rne(x0, x1);
            }
            r = make_uint4(o[0], o[1], o[2], o[3]);
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
    const float c1 = g->scale * 1.4426950408889634f;
    if (f16)
        hipLaunchKernelGGL(xattn6_pack_kernel<true>, dim3(g->B * nch), dim3(256), 0, stream, kv16, ldkv, context_mask, (char*)out->K6, (char*)out->V6,
                           out->vbits, g->T, nch, c1);
    else
        hipLaunchKernelGGL(xattn6_pack_kernel<false>, dim3(g->B * nch), dim3(256), 0, stream, kv16, ldkv, context_mask, (char*)out->K6, (char*)out->V6,
                           out->vbits, g->T, nch, c1);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_xattn6_fwd(const amdnuwa_xattn_geom* g, const uint16_t* q16, int ldq, const amdnuwa_xattn6_kv* kv, const float* null_k,
                                  const float* null_v, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, float* stats,
                                  int f16, hipStream_t stream) {
    int rc = check6(g);
    if (rc) return rc;
    if (!q16 || !kv || !kv->K6 || !kv->V6 || !kv->vbits || !null_k || !null_v || !w_th || !o || ldq % 8 || ldo % 4) return AMDNUWA_ERR_ARG;
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

#if X6_TIMING
extern "C" int amdnuwa_xattn6_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x6_stamps), &p, sizeof(p)); }
#endif

AMDNUWA_SAT_ACCESSOR(xattn6)
