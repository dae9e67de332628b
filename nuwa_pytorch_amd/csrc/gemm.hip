// bf16 MFMA GEMM family for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
//   gemm_nt : C[M,N]  = alpha * A[M,K] . B[N,K]^T (+ bias[N])      every nn.Linear forward
//                                                                   (np.py:274-277, 311-313, 401-405, 1819)
//                                                                   and, with pre-transposed weights, every dgrad
//   gemm_tn : C[N1,N2] = beta*C + alpha * A[M,N1]^T . B[M,N2]        every weight gradient (reduction over tokens)
//
// Both take an optional bf16 "lo" residual per operand (hi+lo split of an fp32 value): with lo
// operands present the kernel issues 3 MFMAs per product (hi*hi + hi*lo + lo*hi) = parity mode.
//
// Token shift (ShiftVideoTokens, np.py:185-253) is folded into the operand LOADER: when
// shift_ntok > 0, logical row g / feature k of the shifted operand is fetched from row g-fmap
// (first quarter of the features, zero at y==0), row g-1 (second quarter, zero at w==0) or row g.
//
// Tile: 128x128 output per 256-thread workgroup (2x2 waves of 64x64 = 4x4 MFMA fragments),
// K-step 64 (NT) / 32 token rows (TN), LDS double-buffered with the next tile's global loads
// in flight during the MFMAs; 16-byte chunks XOR-swizzled so ds_read_b128 / ds_read_b64_tr_b16
// fragment reads are bank-conflict free.
#include <type_traits>
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

struct GemmArgs {
    const bf16_t* A; const bf16_t* Alo; long long sA; int lda;
    const bf16_t* B; const bf16_t* Blo; long long sB; int ldb;
    void* C; bf16_t* Clo; long long sC; int ldc;
    const float* bias;
    float alpha, beta;
    int M, N, K;            // NT: C[M,N], reduce over K.  TN: C[N1=M... see kernel]
    int shift_ntok, shift_fmap, shift_dim;
    int tiles_m, tiles_n;
    int ksplit_len;         // TN: token rows per split
    int batch_inner;        // >0: batch index z -> (z / batch_inner) * stride + (z % batch_inner) * stride_in
    long long sA_in, sB_in, sC_in;
    int nsplit;             // TN 256 ring: number of K splits (the grid is flattened over (split, tile))
    bf16_t* C2; int ldc2;   // NT bf16 epilogue: GEGLU output (C in the interleaved-by-8 layout), or NULL
    bf16_t* C2lo;           // its lo part (bf16x3 ring only)
    int lo_f16;             // bf16x3 ring, EPI 1: Clo receives fp16(value) instead of the bf16 residual
    const bf16_t* Uin; int ldu;   // != NULL: GEGLU BACKWARD epilogue: C2 = du from (product = dgg, Uin = u); C is not written
    int dbg;                // probe only (tuning key 7): bit0 skip epilogue stores, bit1 skip main loop
    // fused linear + cross entropy (EPI 2 / 3 of the 256x256 ring): per (row, 64-column block) partial (max, sum exp) and the
    // target logit in pass 1; dlogits = (exp(logit - lse) - onehot) * ce_scale in pass 2.  The logits never reach memory.
    float* ce_stats; float* ce_tl; const float* ce_lse; const long long* ce_tgt; float ce_scale; int ce_nblk;
    int skew;               // start-phase step of the first-generation workgroups in s_sleep(8) units (tuning key 14; 0 = off)
    int c_f16;              // F16 rings, EPI 1: C / the GEGLU-backward output C2 are fp16 (saturating), not bf16 (the fp16-gradient backward)
    int a_chunk;            // TN whole-M kernel: A is stored as planes of 32 columns, [M / 32][K token rows][32] (amdnuwa_gemm_desc.a_chunk32)
    const float* alpha_dev; // TN whole-M kernel writing C directly: a DEVICE factor on top of alpha (1 / S of the fp16-gradient backward), or NULL
};

__device__ __forceinline__ long long boff(const GemmArgs& p, long long z, long long s, long long s_in) {
    return p.batch_inner > 0 ? (z / p.batch_inner) * s + (z % p.batch_inner) * s_in : z * s;
}

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// source-row offset for the token shift. returns 0 (no shift), a negative offset, or INT_MIN (zero fill)
struct ShiftRow { int off_h, off_w; };
__device__ __forceinline__ ShiftRow shift_row(long long g, int ntok, int fmap) {
    ShiftRow s; s.off_h = 0; s.off_w = 0;
    const int i = (int)(g % ntok);
    if (i == 0) return s;                     // <bos> row is never shifted
    const int p = i - 1;
    const int w = p % fmap, y = (p / fmap) % fmap;
    s.off_h = (y > 0) ? -fmap : INT_MIN;
    s.off_w = (w > 0) ? -1 : INT_MIN;
    return s;
}
__device__ __forceinline__ int shift_pick(const ShiftRow& s, int feat, int quarter) {
    const int q = feat / quarter;
    return q == 0 ? s.off_h : (q == 1 ? s.off_w : 0);
}

// ---------------------------------------------------------------------------------------------
// NT
// ---------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB per operand tile

// byte offset of 16-byte chunk `cc` (0..7) of row `row` inside a [128][64] bf16 tile
__device__ __forceinline__ int nt_lds_off(int row, int cc) { return row * 128 + ((cc ^ ((row >> 1) & 7)) << 4); }

template <bool X3, bool SHIFT, int EPI>   // EPI 0: fp32 out (+bias), 1: bf16 out (hi [+lo])
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf][A_hi, B_hi, (A_lo, B_lo)]
    constexpr int NT_ = X3 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const long long bz = blockIdx.y;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in), oC = boff(p, bz, p.sC, p.sC_in);
    const bf16_t* A = p.A + oA;
    const bf16_t* B = p.B + oB;
    const bf16_t* Alo = X3 ? p.Alo + oA : nullptr;
    const bf16_t* Blo = X3 ? p.Blo + oB : nullptr;

    // loader geometry: 4 chunks per thread per operand; rows (tid>>3) + 32*i, chunk tid&7
    const int lrow = tid >> 3, lcc = tid & 7;
    ShiftRow srow[4];
    if (SHIFT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) srow[i] = shift_row((long long)m0 + lrow + 32 * i, p.shift_ntok, p.shift_fmap);
    }
    const int quarter = SHIFT ? (p.shift_dim >> 2) : 1;

    uint4 ra[4], rb[4], ral[4], rbl[4];
    auto load_tile = [&](int k0) {
        const int k = k0 + lcc * 8;
        const bool kin = k < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i;
            long long ga = (long long)m0 + r;
            bool oka = kin && ga < p.M;
            if (SHIFT && oka) {
                const int off = shift_pick(srow[i], k, quarter);
                if (off == INT_MIN) oka = false; else ga += off;
            }
            ra[i] = oka ? ldg16(A + ga * p.lda + k) : make_uint4(0, 0, 0, 0);
            if (X3) ral[i] = oka ? ldg16(Alo + ga * p.lda + k) : make_uint4(0, 0, 0, 0);
            const long long gb = (long long)n0 + r;
            const bool okb = kin && gb < p.N;
            rb[i] = okb ? ldg16(B + gb * p.ldb + k) : make_uint4(0, 0, 0, 0);
            if (X3) rbl[i] = okb ? ldg16(Blo + gb * p.ldb + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * NT_ * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = nt_lds_off(lrow + 32 * i, lcc);
            *reinterpret_cast<uint4*>(base + off) = ra[i];
            *reinterpret_cast<uint4*>(base + TILE_BYTES + off) = rb[i];
            if (X3) {
                *reinterpret_cast<uint4*>(base + 2 * TILE_BYTES + off) = ral[i];
                *reinterpret_cast<uint4*>(base + 3 * TILE_BYTES + off) = rbl[i];
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const char* base = smem + cur * NT_ * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4], afl[4], bfl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra_ = wm * 64 + i * 16 + fr;
                const int rb_ = wn * 64 + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(base + nt_lds_off(ra_, ks * 4 + fg));
                bfr[i] = *reinterpret_cast<const bf16x8*>(base + TILE_BYTES + nt_lds_off(rb_, ks * 4 + fg));
                if (X3) {
                    afl[i] = *reinterpret_cast<const bf16x8*>(base + 2 * TILE_BYTES + nt_lds_off(ra_, ks * 4 + fg));
                    bfl[i] = *reinterpret_cast<const bf16x8*>(base + 3 * TILE_BYTES + nt_lds_off(rb_, ks * 4 + fg));
                }
            }
            // operands swapped (B as the MFMA "A"): D[row = n][col = m] so each lane ends up with
            // 4 consecutive n of one output row m -> one vector store per fragment.
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (X3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[j], af[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], afl[i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // epilogue
    const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = (long long)m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] * p.alpha;
                if (p.bias && n + r < p.N) v[r] += p.bias[n + r];
            }
            if (EPI == 0) {
                float* C = reinterpret_cast<float*>(p.C) + oC + m * p.ldc + n;
                if (vec_ok) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = v[r];
            } else {
                bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + oC + m * p.ldc + n;
                bf16_t h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) f2bf_hilo(v[r], h[r], l[r]);
                if (vec_ok) {
                    *reinterpret_cast<uint2*>(C) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
                    if (p.Clo) *reinterpret_cast<uint2*>(p.Clo + oC + m * p.ldc + n) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
                } else {
                    for (int r = 0; r < 4 && n + r < p.N; ++r) {
                        C[r] = h[r];
                        if (p.Clo) p.Clo[oC + m * p.ldc + n + r] = l[r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// NT, direct-to-LDS variant: operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B per
// lane, no VGPR staging, no ds_write pass).  The DMA writes LDS lane-linearly (wave-uniform base +
// lane*16), so the bank-conflict swizzle is applied on the per-lane SOURCE address and again on the
// fragment read (same involution).  Zero fill (rows past M/N, shifted rows at y==0 / w==0) comes from
// a 16-byte zero page in global memory.  Smaller LDS footprint (BK=32: 32 KiB) -> 4-5 resident
// workgroups per CU to hide the LDS->MFMA latency.  Requires K % BK == 0; bf16 operands only.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const uint4 g_zero_page[4] = {};

template <int BK_>
__device__ __forceinline__ int glds_swz(int row) {
    if (BK_ == 64) return (row >> 1) & 7;
    // BK 32: 64-byte rows, 4 chunks; row groups of 4 share a bank phase -> permute per group
    const int gsel = (row >> 2) & 3;
    return (0x1320 >> (gsel * 4)) & 3;                     // {0, 2, 3, 1}
}
template <int BK_>
__device__ __forceinline__ int glds_off(int row, int cc) { return row * (BK_ * 2) + ((cc ^ glds_swz<BK_>(row)) << 4); }

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef __attribute__((address_space(1))) const void* glb_cvptr;

template <int BK_, bool SHIFT, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 128 * BK_ * 2;            // bytes per operand tile
    constexpr int NJ = TB / 4096;                // wave-instructions per operand tile per wave (4 waves x 1 KiB)
    constexpr int RPK = 1024 / (BK_ * 2);        // tile rows per KiB
    constexpr int CPR = BK_ / 8;                 // 16-byte chunks per row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const long long bz = blockIdx.y;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in), oC = boff(p, bz, p.sC, p.sC_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);

    // per-lane source pointers (k = 0) for each of this wave's NJ DMA pieces of A and B
    const bf16_t* pa[NJ]; const bf16_t* pah[NJ]; const bf16_t* paw[NJ]; const bf16_t* pb[NJ];
    int cca[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = (j * 4 + wave) * RPK + lane / CPR;
        const int cc = (lane % CPR) ^ glds_swz<BK_>(row);
        cca[j] = cc;
        const long long ga = (long long)m0 + row, gb = (long long)n0 + row;
        pa[j] = ga < p.M ? p.A + oA + ga * p.lda + cc * 8 : nullptr;
        pb[j] = gb < p.N ? p.B + oB + gb * p.ldb + cc * 8 : nullptr;
        pah[j] = paw[j] = pa[j];
        if (SHIFT && pa[j]) {
            const ShiftRow s = shift_row(ga, p.shift_ntok, p.shift_fmap);
            pah[j] = s.off_h == INT_MIN ? nullptr : pa[j] + (long long)s.off_h * p.lda;
            paw[j] = s.off_w == INT_MIN ? nullptr : pa[j] + (long long)s.off_w * p.lda;
        }
    }
    const int quarter = SHIFT ? (p.shift_dim >> 2) : 1;

    auto issue = [&](int buf, int k0) {
        char* base = smem + buf * 2 * TB;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bf16_t* sa = pa[j];
            if (SHIFT) {
                const int kq = k0 + cca[j] * 8, q = (kq >= quarter) + (kq >= 2 * quarter);   // >= 2 -> unshifted half
                sa = q == 0 ? pah[j] : (q == 1 ? paw[j] : pa[j]);
            }
            const bf16_t* srca = sa ? sa + k0 : zp;
            const bf16_t* srcb = pb[j] ? pb[j] + k0 : zp;
            const int off = (j * 4 + wave) * 1024;
            __builtin_amdgcn_global_load_lds((glb_cvptr)srca, (lds_vptr)(base + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_cvptr)srcb, (lds_vptr)(base + TB + off), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK_;
    issue(0, 0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1, (kt + 1) * BK_);
        const char* base = smem + cur * 2 * TB;
#pragma unroll
        for (int ks = 0; ks < BK_ / 32; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(base + glds_off<BK_>(wm * 64 + i * 16 + fr, ks * 4 + fg));
                bfr[i] = *reinterpret_cast<const bf16x8*>(base + TB + glds_off<BK_>(wn * 64 + i * 16 + fr, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = (long long)m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] * p.alpha;
                if (p.bias && n + r < p.N) v[r] += p.bias[n + r];
            }
            if (EPI == 0) {
                float* C = reinterpret_cast<float*>(p.C) + oC + m * p.ldc + n;
                if (vec_ok) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = v[r];
            } else {
                bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + oC + m * p.ldc + n;
                bf16_t h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) f2bf_hilo(v[r], h[r], l[r]);
                if (vec_ok) {
                    *reinterpret_cast<uint2*>(C) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
                    if (p.Clo) *reinterpret_cast<uint2*>(p.Clo + oC + m * p.ldc + n) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
                } else {
                    for (int r = 0; r < 4 && n + r < p.N; ++r) {
                        C[r] = h[r];
                        if (p.Clo) p.Clo[oC + m * p.ldc + n + r] = l[r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// NT, 256x256 tile, 8 waves (2 x 4, 128x64 each), K-step 32, NS-stage direct-to-LDS ring.
// Twice the arithmetic intensity of the 128x128 tile per byte pulled through L2/MALL (128 flop/B), and
// NS-1 K-tiles of DMA in flight: the ring is synchronised with a COUNTED s_waitcnt vmcnt(N) + a raw
// s_barrier per K-tile (a __syncthreads() would drain every outstanding DMA).  Per K-tile per wave:
// 4 global_load_lds, 12 ds_read_b128, 32 MFMAs.  One workgroup per CU (128 KiB LDS, 128 accumulators).
// ---------------------------------------------------------------------------------------------
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// WNW = waves along N: 4 -> 256x256 tile, 512 threads, one workgroup per CU;  2 -> 256x128 tile, 256 threads, 72 KiB of LDS, so
// TWO workgroups share a CU and one's epilogue / pipeline fill overlaps the other's main loop.
// STAG: the two wave rows (waves 0-3 / 4-7 = one wave of each per SIMD) run ONE PHASE APART: every K-step is a read phase
// (12 ds_read_b128 + the DMA issue for a later tile + the counted vmcnt wait + lgkmcnt(0)) and an MFMA phase (32 MFMAs under
// s_setprio 1), separated by raw s_barriers; the lagging row takes one extra barrier up front, so each SIMD always has one
// wave feeding the matrix pipe while its partner pulls the next fragments out of LDS.
// bf16-output instantiations (EPI 1) read their B fragments with PERMUTED rows: MFMA output row rho = 4*fg + r of fragment j is
// tile column fg*16 + j*4 + r, so after the four fragments a lane owns 16 CONTIGUOUS output columns of one row (two 16-byte
// stores, 128 B contiguous per row across the 4 lane groups) instead of four 8-byte pieces.  The B tile then swizzles on
// row bits 4-5 (the bits that vary across a 16-lane ds_read_b128 group under this permutation) to stay bank-conflict free.
template <int EPI>
__device__ __forceinline__ int b256_swz(int row) {
    const int gsel = EPI >= 1 ? (row >> 4) & 3 : (row >> 2) & 3;
    return (0x1320 >> (gsel * 4)) & 3;
}
template <int EPI>
__device__ __forceinline__ int b256_off(int row, int cc) { return row * 64 + ((cc ^ b256_swz<EPI>(row)) << 4); }
template <int EPI>
__device__ __forceinline__ int b256_row(int j, int fr) { return EPI >= 1 ? ((fr >> 2) * 16 + j * 4 + (fr & 3)) : (j * 16 + fr); }

// Fused linear + cross entropy epilogue (to_logits + F.cross_entropy, np.py:1958-1963), shared by the bf16 ring (gemm_nt_256_kernel)
// and the hi + lo ring (gemm_nt_256x3_kernel).  Same ownership as the bf16 epilogue: lane (fr, fg) holds 16 contiguous logits of row
// m; the 4 lanes fr + 16 fg of a row cover this wave's 64-column block (host side guarantees N % 64 == 0).
// EPI 2: per (row, 64-column block) partial (max, sum exp) + the target logit;  EPI 3: dlogits = (exp(logit - lse) - onehot) * ce_scale.
template <int EPI>
__device__ __forceinline__ void ce_epilogue(const GemmArgs& p, const f32x4 (&acc)[8][4], int m0, int n0, int wm, int wn, int fr, int fg) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
        if (m >= p.M) continue;                                   // (the 4 lanes of a row leave together)
        const int nb = n0 + wn * 64 + fg * 16;
        if (nb >= p.N) continue;
        float vv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[j * 4 + r] = acc[i][j][r] * p.alpha;
        const long long t = p.ce_tgt[m];
        const int tc = (t >= nb && t < nb + 16) ? (int)(t - nb) : -1;      // the target column, if this lane holds it
        if constexpr (EPI == 2) {
            float mx = vv[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, vv[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sm = 0.f, tl = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                sm += __builtin_amdgcn_exp2f((vv[e] - mx) * 1.4426950408889634f);
                tl = (e == tc) ? vv[e] : tl;
            }
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            if (fg == 0) *reinterpret_cast<float2*>(p.ce_stats + (m * p.ce_nblk + ((n0 + wn * 64) >> 6)) * 2) = make_float2(mx, sm);
            if (tc >= 0) p.ce_tl[m] = tl;
        } else {
            const float lse = p.ce_lse[m];
            float dv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                dv[e] = (__builtin_amdgcn_exp2f((vv[e] - lse) * 1.4426950408889634f) - (e == tc ? 1.f : 0.f)) * p.ce_scale;
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + nb;
            reinterpret_cast<uint4*>(C)[0] = make_uint4(pack2_rne(dv[0], dv[1]), pack2_rne(dv[2], dv[3]), pack2_rne(dv[4], dv[5]), pack2_rne(dv[6], dv[7]));
            reinterpret_cast<uint4*>(C)[1] = make_uint4(pack2_rne(dv[8], dv[9]), pack2_rne(dv[10], dv[11]), pack2_rne(dv[12], dv[13]), pack2_rne(dv[14], dv[15]));
        }
    }
}

// K-step 64 form of the ring (STAG == 3): operand rows are staged as FULL 128-byte lines (8 rows x 128 B per 1-KiB DMA piece, two
// 64 KiB stages) instead of 64-byte half lines (16 rows x 64 B per piece): half as many L2 requests per byte.  A rows are plain
// (fragment i covers rows i*16 + fr), so the 16-byte chunk swizzle is row & 7; the permuted B rows of the bf16 epilogues
// (b256_row: fr -> (fr >> 2) * 16 + j * 4 + (fr & 3)) vary in row bits 0, 1, 4, 5 inside a fragment, so their swizzle takes bits
// (5, 4, 1): for every ds_read_b128 lane group the 16 addresses then fall on 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int a64_off(int row, int cc) { return row * 128 + ((cc ^ (row & 7)) << 4); }
template <int EPI>
__device__ __forceinline__ int b64_swz(int row) { return EPI >= 1 ? ((((row >> 4) & 3) << 1) | ((row >> 1) & 1)) : (row & 7); }
template <int EPI>
__device__ __forceinline__ int b64_off(int row, int cc) { return row * 128 + ((cc ^ b64_swz<EPI>(row)) << 4); }

// F16: A and B hold fp16 values and the products run on v_mfma_f32_16x16x32_f16 (same rate, 11 significand bits): the FeedForward
// GEMMs of the 'bf16x3-fwd' mode's forward.  Outputs: fp32 (EPI 0) as ever; EPI 1 writes C = bf16 (u, for the bf16 backward) and,
// with C2, the gate output computed on the fp32 accumulators as an fp16 copy (C2: FF2's operand) + a bf16 copy (C2lo: backward).
// Epilogue of the 256-row ring tiles (gemm_nt_256_kernel and its persistent form): accumulators -> C (and the fused second outputs).
// Lane (fr, fg) of wave (wm, wn): row fragments i = 0..7 are rows m0 + wm*128 + i*16 + fr; see the EPI notes at the kernel.
template <int EPI, bool F16, int WNW, bool LT = false>     // LT (EPI 5 on the persistent ring): ltile = a wave-private 4-KiB LDS tile, u comes through coalesced loads
__device__ __forceinline__ void nt256_epilogue(const GemmArgs& p, const f32x4 (&acc)[8][4], int m0, int n0, int wm, int wn, int lane, long long oC,
                                               char* ltile = nullptr) {
    constexpr int BN = 64 * WNW;
    const int fr = lane & 15, fg = lane >> 4;
    const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0);
    if constexpr (EPI == 2 || EPI == 3) {
        ce_epilogue<EPI>(p, acc, m0, n0, wm, wn, fr, fg);
    } else if constexpr (EPI == 1) {
        // lane (fr, fg) owns row m = .. + fr and the 16 contiguous columns n = n0 + wn*64 + fg*16 + [j*4 + r]
        const bool vec8 = (p.N % 8 == 0) && (p.ldc % 8 == 0);
        // the lane's 16 columns are the same for all 8 row fragments: their bias values are loaded ONCE, without a branch.  (Loaded
        // per element inside the row loop each load sat under a branch and was followed by s_waitcnt vmcnt(0) -- which also waits for
        // the STORES of the previous row fragment: 128 dependent round trips per wave and tile on every Linear with a bias.)
        float bias16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bias16[e] = 0.f;
        if (p.bias != nullptr) {                                   // (one uniform branch per tile; no loads at all without a bias)
            const int nbl = n0 + wn * 64 + fg * 16;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = nbl + e < p.N;
                const float bvv = p.bias[in ? nbl + e : 0];
                bias16[e] = in ? bvv : 0.f;
            }
        }
        // GEGLU backward from (product = dgg, u): du = (dgg * gelu(gate), dgg * value * gelu'(gate)) in u's interleaved layout
        auto geglu_bwd_store = [&](const uint4& ua, const uint4& ug, const uint4 (&uu)[4], uint4* dp) {
            const uint4 dg2[2] = {ua, ug};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint4 av = uu[2 * q], gv = uu[2 * q + 1];
                const uint32_t wa[4] = {av.x, av.y, av.z, av.w}, wg[4] = {gv.x, gv.y, gv.z, gv.w};
                const uint32_t wd[4] = {dg2[q].x, dg2[q].y, dg2[q].z, dg2[q].w};
                float da[8], dgt[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y, dy;
                    gelu_both_f(lo_f(wg[k]), y, dy);
                    da[2 * k] = lo_f(wd[k]) * y;
                    dgt[2 * k] = lo_f(wd[k]) * lo_f(wa[k]) * dy;
                    gelu_both_f(hi_f(wg[k]), y, dy);
                    da[2 * k + 1] = hi_f(wd[k]) * y;
                    dgt[2 * k + 1] = hi_f(wd[k]) * hi_f(wa[k]) * dy;
                }
                dp[2 * q] = make_uint4(pack2_rne(da[0], da[1]), pack2_rne(da[2], da[3]), pack2_rne(da[4], da[5]), pack2_rne(da[6], da[7]));
                dp[2 * q + 1] = make_uint4(pack2_rne(dgt[0], dgt[1]), pack2_rne(dgt[2], dgt[3]), pack2_rne(dgt[4], dgt[5]), pack2_rne(dgt[6], dgt[7]));
            }
        };
        if (p.Uin && !p.Clo && vec8 && m0 + 256 <= p.M && n0 + BN <= p.N && !p.dbg) {
            // full tile of the GEGLU backward: a straight-line loop with u of the NEXT row fragment in flight while this one is finished
            // (in the generic loop below every fragment's loads sit behind its row checks and wait with vmcnt(0) -- on the previous
            // fragment's stores as well)
            const int nb = n0 + wn * 64 + fg * 16;
            const bf16_t* ubase = p.Uin + ((long long)m0 + wm * 128 + fr) * p.ldu + 2 * nb;
            bf16_t* dbase = p.C2 + ((long long)m0 + wm * 128 + fr) * p.ldc2 + 2 * nb;
            uint4 un[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) un[q] = reinterpret_cast<const uint4*>(ubase)[q];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint4 uc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) uc[q] = un[q];
                if (i + 1 < 8) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) un[q] = reinterpret_cast<const uint4*>(ubase + (long long)(i + 1) * 16 * p.ldu)[q];
                }
                float vv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[j * 4 + r] = acc[i][j][r] * p.alpha + bias16[j * 4 + r];
                const uint4 ua = make_uint4(pack2_rne(vv[0], vv[1]), pack2_rne(vv[2], vv[3]), pack2_rne(vv[4], vv[5]), pack2_rne(vv[6], vv[7]));
                const uint4 ug = make_uint4(pack2_rne(vv[8], vv[9]), pack2_rne(vv[10], vv[11]), pack2_rne(vv[12], vv[13]), pack2_rne(vv[14], vv[15]));
                geglu_bwd_store(ua, ug, uc, reinterpret_cast<uint4*>(dbase + (long long)i * 16 * p.ldc2));
            }
        } else
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
            if (m >= p.M) continue;
            const int nb = n0 + wn * 64 + fg * 16;
            if (nb >= p.N) continue;
            if ((p.dbg & 1) && acc[i][0][0] != 12345.678f) continue;
            float vv[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[j * 4 + r] = acc[i][j][r] * p.alpha + bias16[j * 4 + r];
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + oC + m * p.ldc + nb;
            bf16_t* Cl = p.Clo ? p.Clo + oC + m * p.ldc + nb : nullptr;
            if ((F16 || !Cl) && vec8 && nb + 16 <= p.N) {          // (F16: Clo, when given, receives the fp16 copy of the product)
                const uint4 ua = make_uint4(pack2_rne(vv[0], vv[1]), pack2_rne(vv[2], vv[3]), pack2_rne(vv[4], vv[5]), pack2_rne(vv[6], vv[7]));
                const uint4 ug = make_uint4(pack2_rne(vv[8], vv[9]), pack2_rne(vv[10], vv[11]), pack2_rne(vv[12], vv[13]), pack2_rne(vv[14], vv[15]));
                if (p.Uin) {
                    // GEGLU backward: the lane's 16 columns of dgg (as bf16, like the stand-alone kernel reads them) meet the two
                    // 16-column groups [8 values | 8 gates] of u they belong to; du leaves in the same interleaved layout
                    const uint4* up = reinterpret_cast<const uint4*>(p.Uin + m * p.ldu + 2 * nb);
                    const uint4 uu[4] = {up[0], up[1], up[2], up[3]};
                    geglu_bwd_store(ua, ug, uu, reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + 2 * nb));
                    continue;
                }
                reinterpret_cast<uint4*>(C)[0] = ua;
                reinterpret_cast<uint4*>(C)[1] = ug;
                if constexpr (F16) {
                    if (Cl) {              // fp16 copy of the product itself (q / k / v for the fp16 attention core)
                        reinterpret_cast<uint4*>(Cl)[0] = make_uint4(pack2_f16_sat(vv[0], vv[1]), pack2_f16_sat(vv[2], vv[3]), pack2_f16_sat(vv[4], vv[5]), pack2_f16_sat(vv[6], vv[7]));
                        reinterpret_cast<uint4*>(Cl)[1] = make_uint4(pack2_f16_sat(vv[8], vv[9]), pack2_f16_sat(vv[10], vv[11]), pack2_f16_sat(vv[12], vv[13]), pack2_f16_sat(vv[14], vv[15]));
                    }
                    if (p.C2) {            // gate on the fp32 accumulators; fp16 copy -> C2 (FF2's A operand), bf16 copy -> C2lo (backward)
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = vv[e] * gelu_f(vv[8 + e]);
                        *reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + (nb >> 1)) =
                            make_uint4(pack2_f16_sat(o[0], o[1]), pack2_f16_sat(o[2], o[3]), pack2_f16_sat(o[4], o[5]), pack2_f16_sat(o[6], o[7]));
                        if (p.C2lo)
                            *reinterpret_cast<uint4*>(p.C2lo + m * p.ldc2 + (nb >> 1)) =
                                make_uint4(pack2_rne(o[0], o[1]), pack2_rne(o[2], o[3]), pack2_rne(o[4], o[5]), pack2_rne(o[6], o[7]));
                    }
                    continue;
                }
                if (p.C2) {
                    // GEGLU on the values as STORED (bf16-rounded u), so the result equals the separate kernel's bit for bit:
                    // the lane's 16 columns are 8 values and their 8 gates (interleaved-by-8 weight rows)
                    const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wg[4] = {ug.x, ug.y, ug.z, ug.w};
                    float o[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o[2 * k] = lo_f(wa[k]) * gelu_f(lo_f(wg[k]));
                        o[2 * k + 1] = hi_f(wa[k]) * gelu_f(hi_f(wg[k]));
                    }
                    *reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + (nb >> 1)) =
                        make_uint4(pack2_rne(o[0], o[1]), pack2_rne(o[2], o[3]), pack2_rne(o[4], o[5]), pack2_rne(o[6], o[7]));
                }
                continue;
            }
            bf16_t h[16], l[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) f2bf_hilo(vv[e], h[e], l[e]);
            if (vec8 && nb + 16 <= p.N) {
                reinterpret_cast<uint4*>(C)[0] = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
                reinterpret_cast<uint4*>(C)[1] = make_uint4(pack2(h[8], h[9]), pack2(h[10], h[11]), pack2(h[12], h[13]), pack2(h[14], h[15]));
                if (Cl) {
                    reinterpret_cast<uint4*>(Cl)[0] = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
                    reinterpret_cast<uint4*>(Cl)[1] = make_uint4(pack2(l[8], l[9]), pack2(l[10], l[11]), pack2(l[12], l[13]), pack2(l[14], l[15]));
                }
            } else {
                for (int e = 0; e < 16 && nb + e < p.N; ++e) {
                    C[e] = h[e];
                    if (Cl) Cl[e] = l[e];
                }
            }
        }
    } else if constexpr (EPI == 4 || EPI == 5) {
        // fp16-gradient backward (round 5; F16 rings only): the product's operands are fp16(S * gradient) and an fp16 weight; ONE fp16 output,
        // saturating.  EPI 4: C itself (dgrad).  EPI 5: du = the GEGLU backward of (product = S dgg, u) in u's interleaved layout (C2; u is
        // bf16 and read element-wise; C is not written).  Separate EPIs so that the forward epilogues (EPI 1) and the long-K kernel (EPI 4
        // only) keep their register budgets: as run-time branches of EPI 1 they cost gemm_nt_256p_kernel 616 B of scratch per lane.
        const int nb = n0 + wn * 64 + fg * 16;
        float sat = 0.f;
        if constexpr (EPI == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
                if (m >= p.M || nb >= p.N) continue;               // (host side: N % 16 == 0, so a lane's 16 columns exist together)
                float vv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[j * 4 + r] = acc[i][j][r] * p.alpha;
                bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + oC + m * p.ldc + nb;
                reinterpret_cast<uint4*>(C)[0] = make_uint4(pack2_f16_sat_n(vv[0], vv[1], sat), pack2_f16_sat_n(vv[2], vv[3], sat), pack2_f16_sat_n(vv[4], vv[5], sat), pack2_f16_sat_n(vv[6], vv[7], sat));
                reinterpret_cast<uint4*>(C)[1] = make_uint4(pack2_f16_sat_n(vv[8], vv[9], sat), pack2_f16_sat_n(vv[10], vv[11], sat), pack2_f16_sat_n(vv[12], vv[13], sat), pack2_f16_sat_n(vv[14], vv[15], sat));
            }
        } else {
            auto geglu_bwd_store16 = [&](const float (&dgg)[16], const uint4 (&uu)[4], uint4* dp) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const uint4 av = uu[2 * q], gv = uu[2 * q + 1];
                    const uint32_t wa[4] = {av.x, av.y, av.z, av.w}, wg[4] = {gv.x, gv.y, gv.z, gv.w};
                    float da[8], dgt[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float y, dy;
                        gelu_both_f(lo_f(wg[k]), y, dy);
                        da[2 * k] = dgg[8 * q + 2 * k] * y;
                        dgt[2 * k] = dgg[8 * q + 2 * k] * lo_f(wa[k]) * dy;
                        gelu_both_f(hi_f(wg[k]), y, dy);
                        da[2 * k + 1] = dgg[8 * q + 2 * k + 1] * y;
                        dgt[2 * k + 1] = dgg[8 * q + 2 * k + 1] * hi_f(wa[k]) * dy;
                    }
                    dp[2 * q] = make_uint4(pack2_f16_sat_n(da[0], da[1], sat), pack2_f16_sat_n(da[2], da[3], sat), pack2_f16_sat_n(da[4], da[5], sat), pack2_f16_sat_n(da[6], da[7], sat));
                    dp[2 * q + 1] = make_uint4(pack2_f16_sat_n(dgt[0], dgt[1], sat), pack2_f16_sat_n(dgt[2], dgt[3], sat), pack2_f16_sat_n(dgt[4], dgt[5], sat), pack2_f16_sat_n(dgt[6], dgt[7], sat));
                }
            };
            if (LT && m0 + 256 <= p.M && n0 + BN <= p.N) {
                // Full tile on the persistent ring (round 6): u arrives through COALESCED loads and a turn in LDS.  A lane owns row fr and the 64 bytes
                // of u behind its 16 columns, so in the direct form below each 16-byte load instruction touches 64 different 64-byte segments.  Here
                // instruction q of a fragment reads rows 4q .. 4q + 3 of the wave's [16 rows][256 B] block, 16 lanes per row (1 KiB contiguous per 4
                // rows), the block is written to the wave's LDS tile as it is consumed and every lane reads its own four pieces back.  Piece p of row
                // r sits at 16-byte slot p ^ g(r), g(r) = r ^ ((r & 4) << 1): the ds_write_b128 / ds_read_b128 lane groups of both directions then
                // touch 16 different slots.  Two fragments are in flight behind the one being consumed, in three NAMED register sets (a modulo-
                // indexed array of them was kept in scratch by the compiler).  tools/geglu_bwd_probe.py, b = 128: 1302 -> 1254 us.
                const int lq = lane >> 4, lp = lane & 15;
                const bf16_t* ucb = p.Uin + ((long long)m0 + wm * 128 + lq) * p.ldu + 2 * (n0 + wn * 64) + lp * 8;
                bf16_t* dbase = p.C2 + ((long long)m0 + wm * 128 + fr) * p.ldc2 + 2 * nb;
                int wpos[4], rpos[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * q + lq;
                    wpos[q] = r * 256 + ((lp ^ (r ^ ((r & 4) << 1))) << 4);
                    rpos[q] = fr * 256 + (((4 * fg + q) ^ (fr ^ ((fr & 4) << 1))) << 4);
                }
                uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3, sc0, sc1, sc2, sc3;
                auto uload = [&](int f, uint4& d0, uint4& d1, uint4& d2, uint4& d3) {
                    d0 = *reinterpret_cast<const uint4*>(ucb + (long long)(f * 16) * p.ldu);
                    d1 = *reinterpret_cast<const uint4*>(ucb + (long long)(f * 16 + 4) * p.ldu);
                    d2 = *reinterpret_cast<const uint4*>(ucb + (long long)(f * 16 + 8) * p.ldu);
                    d3 = *reinterpret_cast<const uint4*>(ucb + (long long)(f * 16 + 12) * p.ldu);
                };
                auto ustep = [&](int i, const f32x4 (&ai)[4], uint4& c0, uint4& c1, uint4& c2, uint4& c3, uint4& n0_, uint4& n1_, uint4& n2_, uint4& n3_) {
                    if (i + 2 < 8) uload(i + 2, n0_, n1_, n2_, n3_);
                    *reinterpret_cast<uint4*>(ltile + wpos[0]) = c0; *reinterpret_cast<uint4*>(ltile + wpos[1]) = c1;
                    *reinterpret_cast<uint4*>(ltile + wpos[2]) = c2; *reinterpret_cast<uint4*>(ltile + wpos[3]) = c3;
                    __builtin_amdgcn_wave_barrier();
                    uint4 uc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) uc[q] = *reinterpret_cast<const uint4*>(ltile + rpos[q]);
                    __builtin_amdgcn_wave_barrier();
                    float vv[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[j * 4 + r] = ai[j][r] * p.alpha;
                    geglu_bwd_store16(vv, uc, reinterpret_cast<uint4*>(dbase + (long long)i * 16 * p.ldc2));
                };
                uload(0, sa0, sa1, sa2, sa3); uload(1, sb0, sb1, sb2, sb3);
                ustep(0, acc[0], sa0, sa1, sa2, sa3, sc0, sc1, sc2, sc3);
                ustep(1, acc[1], sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
                ustep(2, acc[2], sc0, sc1, sc2, sc3, sb0, sb1, sb2, sb3);
                ustep(3, acc[3], sa0, sa1, sa2, sa3, sc0, sc1, sc2, sc3);
                ustep(4, acc[4], sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
                ustep(5, acc[5], sc0, sc1, sc2, sc3, sb0, sb1, sb2, sb3);
                ustep(6, acc[6], sa0, sa1, sa2, sa3, sc0, sc1, sc2, sc3);
                ustep(7, acc[7], sb0, sb1, sb2, sb3, sa0, sa1, sa2, sa3);
            } else if (!LT && m0 + 256 <= p.M && n0 + BN <= p.N) {
                // full tile, direct form (the one-tile ring): a queue of u row fragments that grows as the accumulator registers of finished
                // fragments come free (three in flight at the start, four from fragment 1 on; one fragment ahead: 1302 us, this: 1273 us on the
                // persistent ring before it took the coalesced form above).  sched_barriers keep the compiler from hoisting the loads into a spill.
                const bf16_t* ubase = p.Uin + ((long long)m0 + wm * 128 + fr) * p.ldu + 2 * nb;
                bf16_t* dbase = p.C2 + ((long long)m0 + wm * 128 + fr) * p.ldc2 + 2 * nb;
                uint4 ub[8][4];
                auto uload = [&](int f) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ub[f][q] = reinterpret_cast<const uint4*>(ubase + (long long)f * 16 * p.ldu)[q];
                };
                uload(0); uload(1); uload(2);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 1) { uload(3); uload(4); }
                    if (i >= 2 && i + 3 < 8) uload(i + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    float vv[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[j * 4 + r] = acc[i][j][r] * p.alpha;
                    geglu_bwd_store16(vv, ub[i], reinterpret_cast<uint4*>(dbase + (long long)i * 16 * p.ldc2));
                }
            } else
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
                if (m >= p.M || nb >= p.N) continue;
                float vv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[j * 4 + r] = acc[i][j][r] * p.alpha;
                const uint4* up = reinterpret_cast<const uint4*>(p.Uin + m * p.ldu + 2 * nb);
                const uint4 uu[4] = {up[0], up[1], up[2], up[3]};
                geglu_bwd_store16(vv, uu, reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + 2 * nb));
            }
        }
        f16_sat_commit(sat);
    } else {
    float biasf[4][4];                               // (loaded once per tile: see the bf16 epilogue)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) biasf[j][r] = 0.f;
    if (p.bias != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + j * 16 + fg * 4 + r;
                const bool in = n < p.N;
                const float bvv = p.bias[in ? n : 0];
                biasf[j][r] = in ? bvv : 0.f;
            }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fg * 4;
            if (n >= p.N) continue;
            if ((p.dbg & 1) && acc[i][j][0] != 12345.678f) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha + biasf[j][r];
            float* C = reinterpret_cast<float*>(p.C) + oC + m * p.ldc + n;
            if (vec_ok) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = v[r];
        }
    }
    }
}

template <bool SHIFT, int EPI, int NS, int WNW, int STAG = 0, bool F16 = false>
__global__ __launch_bounds__(WNW * 128, WNW == 2 ? 2 : 1) void gemm_nt_256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 2 * WNW, BN = 64 * WNW;
    constexpr int PA = 16 / NW, PB = (BN / 16) / NW;      // 1 KiB DMA pieces (16 rows x 32 k) per wave: A, B
    constexpr int TB = 256 * 32 * 2;             // A tile: 16 KiB per stage
    constexpr int STG = TB + BN * 32 * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
    const int m0 = tm * 256, n0 = tn * BN;
    const long long bz = blockIdx.y;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in), oC = boff(p, bz, p.sC, p.sC_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);

    const bf16_t* pa[PA]; const bf16_t* pah[PA]; const bf16_t* paw[PA]; const bf16_t* pb[PB];
    int cca[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (j * NW + wave) * 16 + (lane >> 2);
        const int cc = (lane & 3) ^ glds_swz<32>(row);
        cca[j] = cc;
        const long long ga = (long long)m0 + row;
        pa[j] = ga < p.M ? p.A + oA + ga * p.lda + cc * 8 : nullptr;
        pah[j] = paw[j] = pa[j];
        if (SHIFT && pa[j]) {
            const ShiftRow s = shift_row(ga, p.shift_ntok, p.shift_fmap);
            pah[j] = s.off_h == INT_MIN ? nullptr : pa[j] + (long long)s.off_h * p.lda;
            paw[j] = s.off_w == INT_MIN ? nullptr : pa[j] + (long long)s.off_w * p.lda;
        }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = (j * NW + wave) * 16 + (lane >> 2);
        const int cc = (lane & 3) ^ b256_swz<EPI>(row);
        const long long gb = (long long)n0 + row;
        pb[j] = gb < p.N ? p.B + oB + gb * p.ldb + cc * 8 : nullptr;
    }
    const int quarter = SHIFT ? (p.shift_dim >> 2) : 1;
    auto issue_a = [&](int j, int slot, int k0) {
        const bf16_t* sa = pa[j];
        if (SHIFT) {
            const int kq = k0 + cca[j] * 8, q = (kq >= quarter) + (kq >= 2 * quarter);   // >= 2 -> unshifted half
            sa = q == 0 ? pah[j] : (q == 1 ? paw[j] : pa[j]);
        }
        const bf16_t* srca = sa ? sa + k0 : zp;
        __builtin_amdgcn_global_load_lds((glb_cvptr)srca, (lds_vptr)(smem + slot * STG + (j * NW + wave) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int j, int slot, int k0) {
        const bf16_t* srcb = pb[j] ? pb[j] + k0 : zp;
        __builtin_amdgcn_global_load_lds((glb_cvptr)srcb, (lds_vptr)(smem + slot * STG + TB + (j * NW + wave) * 1024), 16, 0, 0);
    };
    auto issue = [&](int slot, int k0) {
#pragma unroll
        for (int j = 0; j < PA; ++j) issue_a(j, slot, k0);
#pragma unroll
        for (int j = 0; j < PB; ++j) issue_b(j, slot, k0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.dbg & 2) ? 0 : p.K / 32;
    // De-phasing.  Every workgroup of the first generation starts at t = 0, reaches its epilogue at the same moment as all the others
    // and the whole chip then stores at once (HBM-bound burst) while the matrix pipes idle, tile after tile.  A start delay spread over
    // 8 phases for the first-generation workgroups (one / two per CU) spreads the epilogues over the tile period: at any moment
    // some CUs store while the others feed the matrix pipe.  p.skew = phase step in s_sleep(8) units (~0.25 us); 0 = off.
    if (p.skew > 0 && (int)blockIdx.x < (WNW == 2 ? 512 : 256)) {
        const int phase = (int)((blockIdx.x * 2654435761u) >> 29);
        for (int i = 0; i < phase * p.skew; ++i) __builtin_amdgcn_s_sleep(8);
    }
    // probe (two workgroups per CU, negative tuning key 14): the SECOND workgroup of every CU (blocks 256..511 of the first generation) starts
    // -skew x 0.25 us late, so that one workgroup's epilogue falls under the other's main loop
    if (WNW == 2 && p.skew < 0 && (int)blockIdx.x >= 256 && (int)blockIdx.x < 512)
        for (int i = 0; i < -p.skew; ++i) __builtin_amdgcn_s_sleep(8);
    // prologue: NS-1 tiles in flight (the K-64 form, STAG == 3, runs its own prologue: NS == 1 here)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s * 32);
    const int fr = lane & 15, fg = lane >> 4;
    if constexpr (STAG == 3 || STAG == 4) {
        // (nothing of the K-32 prologue was issued: NS == 1 for these forms)
        static_assert(WNW == 4 && !SHIFT && NS == 1, "K-64 forms: 8 waves, plain loader");
        constexpr int TB6 = 256 * 64 * 2, STG6 = 2 * TB6;
        const bf16_t* qa[4]; const bf16_t* qb[4];
        int ka[4], kb[4];                                            // k offset (elements) of the lane's 16-byte chunk inside a 64-wide K tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (j * 8 + wave) * 8 + (lane >> 3), slot = lane & 7;
            ka[j] = (slot ^ (row & 7)) * 8;
            kb[j] = (slot ^ b64_swz<EPI>(row)) * 8;
            const long long ga = (long long)m0 + row, gb = (long long)n0 + row;
            qa[j] = ga < p.M ? p.A + oA + ga * p.lda + ka[j] : nullptr;
            qb[j] = gb < p.N ? p.B + oB + gb * p.ldb + kb[j] : nullptr;
        }
        auto issue_a6 = [&](int j, int slot, int k0) {
            const bf16_t* src = (qa[j] && k0 + ka[j] < p.K) ? qa[j] + k0 : zp;
            __builtin_amdgcn_global_load_lds((glb_cvptr)src, (lds_vptr)(smem + slot * STG6 + (j * 8 + wave) * 1024), 16, 0, 0);
        };
        auto issue_b6 = [&](int j, int slot, int k0) {
            const bf16_t* src = (qb[j] && k0 + kb[j] < p.K) ? qb[j] + k0 : zp;
            __builtin_amdgcn_global_load_lds((glb_cvptr)src, (lds_vptr)(smem + slot * STG6 + TB6 + (j * 8 + wave) * 1024), 16, 0, 0);
        };
        const int nk6 = (p.dbg & 2) ? 0 : (p.K + 63) / 64;
        if (nk6 > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) issue_a6(j, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) issue_b6(j, 0, 0);
        }
        if constexpr (STAG == 3) {
        for (int kt = 0; kt < nk6; ++kt) {
            if (kt + 1 < nk6) {                                                  // tile kt has landed (this wave's pieces); kt + 1 in flight
#pragma unroll
                for (int j = 0; j < 4; ++j) issue_a6(j, (kt + 1) & 1, (kt + 1) * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) issue_b6(j, (kt + 1) & 1, (kt + 1) * 64);
                VMCNT(8);
            } else VMCNT(0);
            __builtin_amdgcn_s_barrier();                                        // ... for every wave
            const char* base = smem + (kt & 1) * STG6;
            bf16x8 af[2][8], bfr[2][4];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[s2][j] = *reinterpret_cast<const bf16x8*>(base + TB6 + b64_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), s2 * 4 + fg));
#pragma unroll
                for (int i = 0; i < 8; ++i) af[s2][i] = *reinterpret_cast<const bf16x8*>(base + a64_off(wm * 128 + i * 16 + fr, s2 * 4 + fg));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = mfma16<F16>(bfr[s2][j], af[s2][i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // everyone is done reading stage kt & 1: iteration kt + 1 re-stages it
        }
        } else {
        // STAG == 4: the same two 64 KiB stages with the wave rows ONE PHASE APART, as in the K-step 32 ring: every 32-wide half of a stage
        // is a read phase (12 ds_read_b128 + DMA issue for the NEXT stage) and an MFMA phase (32 MFMAs), raw barriers between them; one row
        // of waves multiplies while the other reads.  The next stage (the slot the lagging row finished reading one barrier ago) is
        // filled during the current one: the leading row issues 6 pieces in its first read phase (the 4 A pieces, which come from furthest
        // away, first) and 2 in its second, and waits for them at the end of its second MFMA phase; the lagging row, whose second read
        // phase IS that phase, issues all 8 in its first read phase.
        VMCNT(0);
        __builtin_amdgcn_s_barrier();                                            // stage 0 has landed for every wave
        if (wm == 1) __builtin_amdgcn_s_barrier();                               // this wave row lags one phase
        for (int st = 0; st < 2 * nk6; ++st) {
            const int kt = st >> 1, s2 = st & 1;
            const char* base = smem + (kt & 1) * STG6;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + TB6 + b64_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), s2 * 4 + fg));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + a64_off(wm * 128 + i * 16 + fr, s2 * 4 + fg));
            if (kt + 1 < nk6) {
                const int ns = (kt + 1) & 1, nk0 = (kt + 1) * 64;
                if (s2 == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) issue_a6(j, ns, nk0);
                    issue_b6(0, ns, nk0); issue_b6(1, ns, nk0);
                    if (wm == 1) { issue_b6(2, ns, nk0); issue_b6(3, ns, nk0); }
                } else if (wm == 0) { issue_b6(2, ns, nk0); issue_b6(3, ns, nk0); }
            }
            if (wm == 1 && s2 == 1) VMCNT(0);                                    // lagging row: its pieces of stage kt + 1 (issued a read phase + an MFMA phase ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma16<F16>(bfr[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (wm == 0 && s2 == 1) VMCNT(0);                                    // leading row: stage kt + 1 complete before the barrier that opens its first read of it
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (STAG == 2) {
        // as STAG 1, but the 4 DMA pieces of the restaged tile are issued INSIDE the MFMA phase, one after every 8 MFMAs (a DMA
        // issue costs ~60 cycles in the shadow of bare MFMAs against 100-185 in a phase that is also pulling fragments out of
        // LDS), so the read phase shrinks to the 12 ds_read_b128.  The leading wave row therefore retires tile kt+1 at the END of
        // its MFMA phase, the lagging row at the end of its read phase -- both one barrier before anyone reads that tile.
        static_assert(PA + PB == 4 && PA == 2 && NS == 4, "written for the 8-wave 4-stage ring");
        if (nk >= 3) VMCNT(8); else if (nk == 2) VMCNT(4); else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const char* base = smem + (kt % NS) * STG;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + TB + b256_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), fg));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + glds_off<32>(wm * 128 + i * 16 + fr, fg));
            const int rem2 = nk - 2 - kt;
            if (wm == 1) { if (rem2 >= 1) VMCNT(4); else VMCNT(0); }      // lagging row: tiles <= kt+2 issued so far
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool more = kt + NS - 1 < nk;
            const int nslot = (kt + NS - 1) % NS, nk0 = (kt + NS - 1) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 2 * q; i < 2 * q + 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = mfma16<F16>(bfr[j], af[i], acc[i][j]);
                if (more) { if (q < 2) issue_a(q, nslot, nk0); else issue_b(q - 2, nslot, nk0); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (wm == 0) { if (rem2 >= 2) VMCNT(8); else if (rem2 == 1) VMCNT(4); else VMCNT(0); }   // leading row: tiles <= kt+3 issued
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (STAG == 1) {
        static_assert(PA + PB == 4 && NS == 4, "staggered schedule is written for the 8-wave 4-stage ring");
        // tile 0 landed (for this wave: at most the later prologue tiles still in flight), then for everyone
        if (nk >= 3) VMCNT(8); else if (nk == 2) VMCNT(4); else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();           // this wave row lags one phase
        for (int kt = 0; kt < nk; ++kt) {
            // ---- read phase: fragments of tile kt; restage the slot tile kt-1 lived in (its last reader finished a phase ago)
            const char* base = smem + (kt % NS) * STG;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + TB + b256_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), fg));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + glds_off<32>(wm * 128 + i * 16 + fr, fg));
            if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, (kt + NS - 1) * 32);
            // tile kt+1 must have landed (this wave's share) before the barrier that precedes anyone's read of it
            const int rem2 = nk - 2 - kt;
            if (rem2 >= 2) VMCNT(8); else if (rem2 == 1) VMCNT(4); else VMCNT(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- MFMA phase
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma16<F16>(bfr[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (p.dbg & 64) {
                // PROBE (tuning key 7 bit 6, with bit 0 = no epilogue): can a tile's worth of output stores drain in the shadow of a main loop?  Every
                // K-step issues 1/16 of the wave's bf16 (+ second copy) tile stores from live accumulator registers to the tile's own output rows
                // (garbage values, real addresses and sizes): 8 rows x 16 K-steps = the 128 row pieces of the epilogue.
                const int i8 = (kt & 7), fr_ = lane & 15, fg_ = lane >> 4;
                const long long m = (long long)m0 + wm * 128 + i8 * 16 + fr_;
                const int nb = n0 + wn * 64 + fg_ * 16;
                if (m < p.M && nb + 16 <= p.N && (kt >> 3) < 2) {
                    bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + oC + m * p.ldc + nb;
                    const uint4 v0 = __builtin_bit_cast(uint4, bfr[0]);          // (any live 16 bytes that no MFMA is writing)
                    if ((kt >> 3) == 0) { reinterpret_cast<uint4*>(C)[0] = v0; reinterpret_cast<uint4*>(C)[1] = v0; }
                    else if (p.Clo) { bf16_t* Cl = p.Clo + oC + m * p.ldc + nb; reinterpret_cast<uint4*>(Cl)[0] = v0; reinterpret_cast<uint4*>(Cl)[1] = v0; }
                    else if (p.C2) {
                        *reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + (nb >> 1)) = v0;
                        if (p.C2lo) *reinterpret_cast<uint4*>(p.C2lo + m * p.ldc2 + (nb >> 1)) = v0;
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    } else
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed for THIS wave once at most (tiles after kt still in flight) x 4 DMAs remain
        const int rem = nk - 1 - kt;
        if constexpr (PA + PB == 4) {
            if (NS >= 4 && rem >= 2) VMCNT(8);
            else if (rem >= 1) VMCNT(4);
            else VMCNT(0);
        } else {
            if (NS >= 4 && rem >= 2) VMCNT(12);
            else if (rem >= 1) VMCNT(6);
            else VMCNT(0);
        }
        __builtin_amdgcn_s_barrier();            // ... and for every wave; also: everyone finished reading slot (kt-1)%NS
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, (kt + NS - 1) * 32);
        const char* base = smem + (kt % NS) * STG;
        bf16x8 af[8], bfr[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + glds_off<32>(wm * 128 + i * 16 + fr, fg));
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + TB + b256_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), fg));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = mfma16<F16>(bfr[j], af[i], acc[i][j]);
    }

    nt256_epilogue<EPI, F16, WNW>(p, acc, m0, n0, wm, wn, lane, oC);
}

// ---------------------------------------------------------------------------------------------
// NT, 256x256 tile, PERSISTENT form of the 8-wave staggered ring (gemm_nt_256_kernel<false, EPI, 4, 4, 1, F16>): one workgroup per CU
// walks the tile list (virtual block ids blockIdx.x, + gridDim.x, ...: the order the hardware would have dispatched them in, so the
// XCD remap keeps its meaning) and the 4-stage DMA ring runs ON across tile borders: the last three K-steps of a tile already stage
// the first three of the next one, so the epilogue's stores leave while those operands are in flight and the next main loop starts on
// landed data.  A one-tile-per-workgroup launch pays, per tile, the ring fill (HBM / L2 latency, about 2 us) and the drain of its own
// stores with an idle matrix pipe -- on the K = 512 products a quarter of the 7 us main loop.  (The vendor library's kernels for
// these shapes are persistent stream-K kernels; that is where the idea comes from.)
// Memory operations retire in order, so a counted vmcnt wait for a DMA group also covers everything older: right after an epilogue
// the waits of K-steps 0 and 1 (whose groups were issued BEFORE the stores) allow the epilogue's store count on top of the two younger
// DMA groups; from K-step 2 on the stores are older than the group waited for and have drained with it.  The store count is exact for
// full tiles of the plain epilogues and taken as 0 (= wait for everything, always safe) otherwise.
// Results are bit-identical to the one-tile kernel (same accumulation order).  Needs K / 32 >= 3.
// ---------------------------------------------------------------------------------------------
template <int EPI, bool F16>
__global__ __launch_bounds__(512, 1) void gemm_nt_256p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 4, NW = 8;
    constexpr int TB = 256 * 32 * 2, STG = 2 * TB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = p.tiles_m * p.tiles_n;
    const long long bz = blockIdx.y;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in), oC = boff(p, bz, p.sC, p.sC_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = p.K / 32;
    int vb = blockIdx.x;
    if (vb >= nwg) return;

    const bf16_t* pa[2]; const bf16_t* pb[2]; const bf16_t* pan[2]; const bf16_t* pbn[2];
    int m0, n0, m0n = 0, n0n = 0;
    auto setup = [&](int v, const bf16_t* (&qa)[2], const bf16_t* (&qb)[2], int& mm, int& nn) {
        const int lid = xcd_remap(v, nwg);
        const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
        mm = tm * 256; nn = tn * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (j * NW + wave) * 16 + (lane >> 2);
            const long long ga = (long long)mm + row, gb = (long long)nn + row;
            qa[j] = ga < p.M ? p.A + oA + ga * p.lda + ((lane & 3) ^ glds_swz<32>(row)) * 8 : nullptr;
            qb[j] = gb < p.N ? p.B + oB + gb * p.ldb + ((lane & 3) ^ b256_swz<EPI>(row)) * 8 : nullptr;
        }
    };
    auto issue = [&](const bf16_t* const (&qa)[2], const bf16_t* const (&qb)[2], int slot, int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((glb_cvptr)(qa[j] ? qa[j] + k0 : zp), (lds_vptr)(smem + slot * STG + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((glb_cvptr)(qb[j] ? qb[j] + k0 : zp), (lds_vptr)(smem + slot * STG + TB + (j * NW + wave) * 1024), 16, 0, 0);
    };
    auto wait_vm = [&](int n) {                                                  // (uniform; rounds DOWN to an immediate: never too permissive)
        if (n >= 40) VMCNT(40); else if (n >= 24) VMCNT(24); else if (n >= 8) VMCNT(8); else if (n >= 4) VMCNT(4); else VMCNT(0);
    };
    // vector-memory instructions every wave issues in the epilogue of a FULL tile (0 = unknown: wait for everything)
    int ns_full = 0;
    if (!p.dbg) {
        if constexpr (EPI == 0) ns_full = (p.N % 4 == 0 && p.ldc % 4 == 0) ? 32 : 0;
        else if constexpr (EPI == 1) {
            if (!p.Uin && p.N % 8 == 0 && p.ldc % 8 == 0 && !p.bias) {
                if (F16) ns_full = 16 + (p.Clo ? 16 : 0) + (p.C2 ? 8 + (p.C2lo ? 8 : 0) : 0);
                else ns_full = p.Clo ? 32 : 16 + (p.C2 ? 8 : 0);
            }
        } else if constexpr (EPI == 4) {
            if (p.N % 16 == 0 && p.ldc % 8 == 0) ns_full = 16;
        }
    }

    setup(vb, pa, pb, m0, n0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(pa, pb, s, s * 32);                   // (nk >= 3)
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    VMCNT(8);
    __builtin_amdgcn_s_barrier();                                                // step 0 of the first tile has landed for every wave
    int slot = 0, ns_prev = 0;                                                   // ns_prev: stores of the epilogue just behind us (this wave)
    for (;;) {
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < nwg;
        if (has_next) setup(vbn, pan, pbn, m0n, n0n);
        if (wm == 1) __builtin_amdgcn_s_barrier();                               // this wave row lags one phase
        for (int kt = 0; kt < nk; ++kt) {
            // ---- read phase: fragments of step kt; restage the slot step kt - 1 lived in with step kt + 3 (of this tile or the next)
            const char* base = smem + slot * STG;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(base + TB + b256_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), fg));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + glds_off<32>(wm * 128 + i * 16 + fr, fg));
            const int tgt = kt + NS - 1, rslot = (slot + NS - 1) & (NS - 1);
            if (tgt < nk) issue(pa, pb, rslot, tgt * 32);
            else if (has_next) issue(pan, pbn, rslot, (tgt - nk) * 32);
            // step kt + 1 (this wave's share) must have landed before the barrier that precedes anyone's read of it: allowed in flight =
            // the younger DMA groups (steps kt + 2, kt + 3 where they exist) + the previous epilogue's stores while they are younger too
            const int g2 = (kt + 2 < nk || has_next) ? 4 : 0, g3 = (kt + 3 < nk || has_next) ? 4 : 0;
            wait_vm(g2 + g3 + (kt <= 1 ? ns_prev : 0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- MFMA phase
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma16<F16>(bfr[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            slot = (slot + 1) & (NS - 1);
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();                               // rows back in step: both run their epilogues together
        nt256_epilogue<EPI, F16, 4, EPI == 5>(p, acc, m0, n0, wm, wn, lane, oC, smem + NS * STG + wave * 4096);   // (EPI 5: 8 x 4 KiB of u tiles behind the ring, see the launcher)
        if (!has_next) break;
        ns_prev = (m0 + 256 <= p.M && n0 + 256 <= p.N) ? ns_full : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        vb = vbn; m0 = m0n; n0 = n0n;
#pragma unroll
        for (int j = 0; j < 2; ++j) { pa[j] = pan[j]; pb[j] = pbn[j]; }
    }
}

// (round 6: gemm_nt_w4_kernel -- the four-wave K-step 32 predecessor of gemm_nt_w4k_kernel, reachable through tuning key 0 = 9 only -- and its
//  epilogue were removed; docs/DESIGN_history_r1-r4.md section 5v describes it)

// ---------------------------------------------------------------------------------------------
// NT, 256x256 tile, bf16x3: every operand a bf16 hi + lo pair, three MFMAs per product (lo*hi + hi*lo + hi*hi, the small
// terms first) -- the arithmetic of the forward in the 'bf16x3' and 'bf16x3-fwd' precision modes (the modes whose logits stay
// within 1e-3 of the fp32 reference).  Same 8-wave 2 x 4 layout, accumulator ownership, B-row permutation and epilogues as the
// bf16 ring above; what differs is the staging: a stage holds FOUR 16 KiB operand images (A hi, A lo, B hi, B lo), so the ring
// is 2 stages deep (128 KiB).  That is enough here: a K-step is 96 MFMAs per wave (~3.2k cycles per SIMD with its two waves)
// against 24 ds_read_b128 and 8 DMA pieces, i.e. the loop is bound by the matrix pipe and the next stage's DMA has a whole
// MFMA phase to land.  DMA pieces go through dma16_asm (common.h): with the builtin the compiler drains the ring (vmcnt(0)) in
// front of the fragment reads that follow the issue in the same basic block.
// Accumulation order per output element (k ascending; per 32-chunk lo*hi, hi*lo, hi*hi) equals gemm_nt_kernel<true,...>'s, so
// the two kernels agree bit for bit.
// ---------------------------------------------------------------------------------------------
// X2: the TWO-MFMA form of the same ring (amdnuwa_gemm_desc.ab_f16 with Blo): A is ONE fp16 image (an activation rounded to 11 significand
// bits), B an fp16 hi + lo pair (the weight to ~22 bits); per 32-chunk lo*a then hi*a on v_mfma_f32_16x16x32_f16.  A stage holds three
// images (A | B hi | B lo, 48 KiB); everything else -- wave layout, B-row permutation, epilogues -- is the three-MFMA kernel's.
// (A third 48 KiB stage fits and was measured: a tie on all three shapes, 407 / 420 / 6441 us with two stages against 411 / 423 / 6486 --
// the loop is not waiting for its DMA.)
template <int EPI, bool X2 = false>      // 0: fp32 out (+bias);  1: bf16 hi [+ lo] out (+bias);  2 / 3: the two passes of the fused cross entropy (ce_epilogue)
__global__ __launch_bounds__(512) void gemm_nt_256x3_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 256 * 32 * 2;             // one operand image of a stage: 16 KiB
    constexpr int STG = (X2 ? 3 : 4) * TB;       // A hi | A lo | B hi | B lo   (X2: A | B hi | B lo)
    constexpr int OBH = (X2 ? 1 : 2) * TB, OBL = OBH + TB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const long long bz = blockIdx.y;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in), oC = boff(p, bz, p.sC, p.sC_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);

    // this wave's DMA pieces: pieces j*8 + wave (j = 0, 1) of each of the four images; a piece = 16 rows x 32 k (1 KiB)
    long long offa[2], offb[2];                  // element offsets of the lane's 16-byte chunk at k = 0; < 0: row outside -> zero page
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 16 + (lane >> 2);
        const int cca = (lane & 3) ^ glds_swz<32>(row);
        const int ccb = (lane & 3) ^ b256_swz<EPI>(row);
        const long long ga = (long long)m0 + row, gb = (long long)n0 + row;
        offa[j] = ga < p.M ? oA + ga * p.lda + cca * 8 : -1;
        offb[j] = gb < p.N ? oB + gb * p.ldb + ccb * 8 : -1;
    }
    auto issue = [&](int slot, int k0) {
        char* base = smem + slot * STG;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int po = (j * 8 + wave) * 1024;
            const bool ina = offa[j] >= 0, inb = offb[j] >= 0;
            dma16_asm(ina ? p.A + offa[j] + k0 : zp, base + po);
            if constexpr (!X2) dma16_asm(ina ? p.Alo + offa[j] + k0 : zp, base + TB + po);
            dma16_asm(inb ? p.B + offb[j] + k0 : zp, base + OBH + po);
            dma16_asm(inb ? p.Blo + offb[j] + k0 : zp, base + OBL + po);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.dbg & 2) ? 0 : p.K / 32;
    const int fr = lane & 15, fg = lane >> 4;
    // start-phase skew of the first-generation workgroups (see gemm_nt_256_kernel): spreads the chip-wide epilogue store bursts
    if (p.skew > 0 && (int)blockIdx.x < 256) {
        const int phase = (int)((blockIdx.x * 2654435761u) >> 29);
        for (int i = 0; i < phase * p.skew; ++i) __builtin_amdgcn_s_sleep(8);
    }
    if (nk > 0) issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        VMCNT(0);                                 // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();             // ... and everyone's; also: everyone is done reading the other slot (tile kt-1)
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * 32);
        const char* base = smem + (kt & 1) * STG;
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = b256_off<EPI>(wn * 64 + b256_row<EPI>(j, fr), fg);
            bh[j] = *reinterpret_cast<const bf16x8*>(base + OBH + o);
            bl[j] = *reinterpret_cast<const bf16x8*>(base + OBL + o);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = glds_off<32>(wm * 128 + (hf * 4 + i) * 16 + fr, fg);
                ah[i] = *reinterpret_cast<const bf16x8*>(base + o);
                if constexpr (!X2) al[i] = *reinterpret_cast<const bf16x8*>(base + TB + o);
            }
            if constexpr (X2) {
                // two sweeps over the 16 accumulators of this half (the small term first), fp16 MFMAs
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[hf * 4 + i][j] = mfma16<true>(bl[j], ah[i], acc[hf * 4 + i][j]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[hf * 4 + i][j] = mfma16<true>(bh[j], ah[i], acc[hf * 4 + i][j]);
                continue;
            }
            // three sweeps over the 16 accumulators of this half: consecutive MFMAs never share an accumulator
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[hf * 4 + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[hf * 4 + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[hf * 4 + i][j], 0, 0, 0);
        }
    }

    if constexpr (EPI == 2 || EPI == 3) {
        ce_epilogue<EPI>(p, acc, m0, n0, wm, wn, fr, fg);        // fused to_logits + cross entropy on the hi + lo product (amdnuwa_linear_ce_x3)
    } else if constexpr (EPI == 1) {
        // lane (fr, fg) owns row m = .. + fr and the 16 contiguous columns n = n0 + wn*64 + fg*16 + [j*4 + r] (permuted B rows)
        const bool vec8 = (p.N % 8 == 0) && (p.ldc % 8 == 0);
        const int nb = n0 + wn * 64 + fg * 16;
        float bias16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool in = p.bias != nullptr && nb + e < p.N;
            const float bvv = (p.bias ? p.bias : reinterpret_cast<const float*>(g_zero_page))[in ? nb + e : 0];
            bias16[e] = in ? bvv : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
            if (m >= p.M || nb >= p.N) continue;
            if ((p.dbg & 1) && acc[i][0][0] != 12345.678f) continue;
            // the lane's 16 values as 8 packed pairs through the gfx950 converters (v_cvt_pk_bf16_f32: round-to-nearest-even, bit-identical
            // to the integer form f2bf_hilo on finite values): hi pair, then EITHER the bf16 residual pair OR (lo_f16, wave-uniform) the fp16
            // rendering -- the first form computed both for all 16 elements with ~20 integer instructions each, which under the build
            // without packed fp32 ops pushed this epilogue 80 B per lane over its register budget
            uint32_t hp[8], lp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e0 = 2 * k, e1 = 2 * k + 1;
                const float v0 = acc[i][e0 >> 2][e0 & 3] * p.alpha + bias16[e0], v1 = acc[i][e1 >> 2][e1 & 3] * p.alpha + bias16[e1];
                hp[k] = p.c_f16 ? pack2_f16_sat(v0, v1) : pack2_rne(v0, v1);      // (c_f16, wave-uniform: C is ONE fp16 output, no second copy)
                lp[k] = p.lo_f16 ? pack2_f16_sat(v0, v1) : pack2_rne(v0 - lo_f(hp[k]), v1 - hi_f(hp[k]));
            }
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + oC + m * p.ldc + nb;
            bf16_t* Cl = p.Clo ? p.Clo + oC + m * p.ldc + nb : nullptr;
            if (vec8 && nb + 16 <= p.N) {
                reinterpret_cast<uint4*>(C)[0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                reinterpret_cast<uint4*>(C)[1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
                if (Cl) {
                    reinterpret_cast<uint4*>(Cl)[0] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
                    reinterpret_cast<uint4*>(Cl)[1] = make_uint4(lp[4], lp[5], lp[6], lp[7]);
                }
                if (p.C2) {
                    // FF1 (np.py:274-277): the lane's 16 columns are 8 values and their 8 gates (interleaved-by-8 weight rows); the gate runs
                    // on the fp32 accumulators themselves and leaves as a hi + lo pair, the A operand of FF2.  u's own lo part is only
                    // written when the caller wants it (the bf16x3 backward); the bf16x3-fwd mode keeps u.hi for its bf16 backward.
                    uint32_t gh[4], gl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float o2[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int e = 2 * k + t;
                            const float av = acc[i][e >> 2][e & 3] * p.alpha + bias16[e];
                            const float gv = acc[i][2 + (e >> 2)][e & 3] * p.alpha + bias16[8 + e];
                            o2[t] = av * gelu_f(gv);
                        }
                        gh[k] = pack2_rne(o2[0], o2[1]);
                        gl[k] = pack2_rne(o2[0] - lo_f(gh[k]), o2[1] - hi_f(gh[k]));
                    }
                    *reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + (nb >> 1)) = make_uint4(gh[0], gh[1], gh[2], gh[3]);
                    if (p.C2lo) *reinterpret_cast<uint4*>(p.C2lo + m * p.ldc2 + (nb >> 1)) = make_uint4(gl[0], gl[1], gl[2], gl[3]);
                }
            } else {
                for (int e = 0; e < 16 && nb + e < p.N; ++e) {
                    C[e] = (bf16_t)((e & 1) ? (hp[e >> 1] >> 16) : (hp[e >> 1] & 0xffffu));
                    if (Cl) Cl[e] = (bf16_t)((e & 1) ? (lp[e >> 1] >> 16) : (lp[e >> 1] & 0xffffu));
                }
            }
        }
    } else {
        const bool vec_ok = (p.N % 4 == 0) && (p.ldc % 4 == 0);
        float biasf[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + j * 16 + fg * 4 + r;
                const bool in = p.bias != nullptr && n < p.N;
                const float bvv = (p.bias ? p.bias : reinterpret_cast<const float*>(g_zero_page))[in ? n : 0];
                biasf[j][r] = in ? bvv : 0.f;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long m = (long long)m0 + wm * 128 + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fg * 4;
                if (n >= p.N) continue;
                if ((p.dbg & 1) && acc[i][j][0] != 12345.678f) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha + biasf[j][r];
                float* C = reinterpret_cast<float*>(p.C) + oC + m * p.ldc + n;
                if (vec_ok) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// NT, 256x256 tile, FOUR waves (2 x 2, 128x128 each), K-step 64, TWO 64 KiB stages with a 1.5-iteration prefetch: the long-K form.
// Structure after the vendor library's kernels for these shapes: a K-step's fragments -- 16 + 16 ds_read_b128, all 128
// registers of them -- are double-buffered in registers, so a stage is free for re-staging as soon as its SECOND half has been read,
// half an iteration before its MFMAs are done:
//   P1   64 MFMAs on half 0 (registers)   |  the 16 reads of half 1 of stage s
//        lgkmcnt(0), barrier                 (stage s consumed by every wave)
//   P2   64 MFMAs on half 1               |  16 DMA pieces of iteration i + 2 -> stage s, then vmcnt + barrier (iteration i + 1 has landed
//                                            in stage s ^ 1), then the 16 reads of ITS half 0
// Operand rows travel as full 128-byte lines (8 rows x 128 B per 1-KiB DMA piece; half the L2 requests of the K-step 32 ring), source
// = a scalar base advanced by 128 bytes per iteration + a constant per-lane byte offset.  Every instruction of the loop is placed by
// hand (sched_barrier between the slots): one MFMA, at most one other instruction, one MFMA ...  K % 32 == 0 (an odd last half is zeroed).
// Accumulation order per output element = k ascending in chunks of 32, as every other NT kernel here: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <int EPI, bool F16>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4k_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 256 * 64 * 2;             // one operand image of a stage: 32 KiB
    constexpr int STG = 2 * TB;                  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const long long bz = blockIdx.y;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in), oC = boff(p, bz, p.sC, p.sC_in);
    const unsigned lds0 = (unsigned)(size_t)(lds_vptr_t)smem;
    const char* baseA = reinterpret_cast<const char*>(p.A + oA + (long long)m0 * p.lda);
    const char* baseB = reinterpret_cast<const char*>(p.B + oB + (long long)n0 * p.ldb);
    // this wave's DMA pieces: pieces j * 4 + wave (j = 0..7) of the A and of the B image; a piece = 8 rows x 64 k (1 KiB): lane -> row
    // lane >> 3, LDS chunk lane & 7, which holds source chunk (lane & 7) ^ swizzle(row).  Rows past M / N are clamped to the last row.
    unsigned offa[8], offb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j * 4 + wave) * 8 + (lane >> 3);
        const int ra = min(row, p.M - 1 - m0), rb = min(row, p.N - 1 - n0);
        offa[j] = (unsigned)((ra * p.lda + (((lane & 7) ^ (row & 7)) * 8)) * 2);
        offb[j] = (unsigned)((rb * p.ldb + (((lane & 7) ^ b64_swz<EPI>(row)) * 8)) * 2);
    }
    // K % 64 == 32 (FF2's 1376; K >= 96): one more iteration, staged from k = K - 64 (so that nothing past the end of a row is read): its
    // first half repeats the previous iteration's second half and is ZEROED in registers before it is multiplied, its second half is the
    // last 32 of K.  The sums are exact (x + 0 * finite) and keep their order.
    const bool odd = (p.K & 63) != 0;
    const int nit = (p.dbg & 2) ? 0 : (p.K + 63) / 64;
    auto issue_piece = [&](int q, int stage, int it) {            // q = 0..7: A pieces, 8..15: B pieces
        const char* sb = (q < 8 ? baseA : baseB) + (size_t)it * 128 - ((odd && it == nit - 1) ? 64 : 0);
        const unsigned long long sbu = (unsigned long long)sb;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sbu), hi = __builtin_amdgcn_readfirstlane((unsigned)(sbu >> 32));
        const unsigned long long sbase = ((unsigned long long)hi << 32) | lo;
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane(lds0 + stage * STG + (q < 8 ? 0 : TB) + ((q & 7) * 4 + wave) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(q < 8 ? offa[q & 7] : offb[q & 7]), "s"(sbase), "s"(lds_addr)
                     : "memory", "m0");
    };

    f32x4 acc[2][8][4];                          // [column half][row fragment][column fragment]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    // fragment addresses (stage 0), per K half h: A fragment i at + i * 2048; B fragment (c, j) at + c * 8192 + j * JB
    constexpr int JB = EPI >= 1 ? 512 : 2048;
    const int browl = EPI >= 1 ? (fr >> 2) * 16 + (fr & 3) : fr;
    const int bswz = EPI >= 1 ? (((fr >> 2) << 1) | ((fr >> 1) & 1)) : (fr & 7);
    unsigned la[2], lb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        la[h] = lds0 + (wm * 128 + fr) * 128 + (((h * 4 + fg) ^ (fr & 7)) << 4);
        lb[h] = lds0 + TB + (wn * 128 + browl) * 128 + (((h * 4 + fg) ^ bswz) << 4);
    }
    bf16x8 af[2][8], bfr[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) af[h][i] = bfr[h][i] = bf16x8{};
#define WK_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(DST) : "v"(ADDR), "n"(OFF))
#define WK_RDA(H, I, ADDR) WK_RD(af[H][I], ADDR, (I) * 2048)
#define WK_RDB(H, Q, ADDR) WK_RD(bfr[H][Q], ADDR, ((Q) >> 2) * 8192 + ((Q) & 3) * JB)
#define WK_PIN() __builtin_amdgcn_sched_barrier(0)
#define WK_MF(H, I, Q) do { acc[(Q) >> 2][I][(Q) & 3] = mfma16<F16>(bfr[H][Q], af[H][I], acc[(Q) >> 2][I][(Q) & 3]); WK_PIN(); } while (0)
    // everything the MFMAs read has to be known to the compiler as written: tie the fragment registers of a half to the wait that retires them
#define WK_LGKM0(H)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[H][0]), "+v"(af[H][1]), "+v"(af[H][2]), "+v"(af[H][3]), "+v"(af[H][4]), "+v"(af[H][5]), \
                 "+v"(af[H][6]), "+v"(af[H][7]), "+v"(bfr[H][0]), "+v"(bfr[H][1]), "+v"(bfr[H][2]), "+v"(bfr[H][3]), "+v"(bfr[H][4]),       \
                 "+v"(bfr[H][5]), "+v"(bfr[H][6]), "+v"(bfr[H][7]) :: "memory")
    if (nit > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) issue_piece(q, 0, 0);
        if (nit > 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) issue_piece(q, 1, 1);
            VMCNT(16);
        } else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        WK_RDB(0, 0, lb[0]); WK_RDB(0, 1, lb[0]); WK_RDB(0, 2, lb[0]); WK_RDB(0, 3, lb[0]); WK_RDB(0, 4, lb[0]); WK_RDB(0, 5, lb[0]); WK_RDB(0, 6, lb[0]); WK_RDB(0, 7, lb[0]);
        WK_RDA(0, 0, la[0]); WK_RDA(0, 1, la[0]); WK_RDA(0, 2, la[0]); WK_RDA(0, 3, la[0]); WK_RDA(0, 4, la[0]); WK_RDA(0, 5, la[0]); WK_RDA(0, 6, la[0]); WK_RDA(0, 7, la[0]);
        WK_LGKM0(0);
    }
    WK_PIN();
    // One branch-free body for every iteration: past the end of K the staging simply repeats the LAST K-step into the stage that has just
    // been consumed (nobody reads it again) and the fragment reads of a non-existent next step fetch values nobody multiplies -- two
    // redundant L2-resident K-steps per tile (the kernel is for long K) instead of 37 scalar branches per iteration.
    for (int it = 0; it < nit; ++it) {
        constexpr bool more2 = true, more1 = true;
        const int it2 = min(it + 2, nit - 1);
        const unsigned so = (unsigned)((it & 1) * STG), sn = (unsigned)(((it + 1) & 1) * STG);
        const unsigned a1 = la[1] + so, b1 = lb[1] + so, a0n = la[0] + sn, b0n = lb[0] + sn;
        // ---- P1: half 0 from registers; the 16 reads of half 1 ride between the first 32 MFMAs
#define WK_ROW_P1(I, R0, R1, R2, R3)                                                                                \
        WK_MF(0, I, 0); R0; WK_PIN(); WK_MF(0, I, 1); WK_MF(0, I, 2); R1; WK_PIN(); WK_MF(0, I, 3);                  \
        WK_MF(0, I, 4); R2; WK_PIN(); WK_MF(0, I, 5); WK_MF(0, I, 6); R3; WK_PIN(); WK_MF(0, I, 7)
        WK_ROW_P1(0, WK_RDB(1, 0, b1), WK_RDB(1, 1, b1), WK_RDB(1, 2, b1), WK_RDB(1, 3, b1));
        WK_ROW_P1(1, WK_RDB(1, 4, b1), WK_RDB(1, 5, b1), WK_RDB(1, 6, b1), WK_RDB(1, 7, b1));
        WK_ROW_P1(2, WK_RDA(1, 0, a1), WK_RDA(1, 1, a1), WK_RDA(1, 2, a1), WK_RDA(1, 3, a1));
        WK_ROW_P1(3, WK_RDA(1, 4, a1), WK_RDA(1, 5, a1), WK_RDA(1, 6, a1), WK_RDA(1, 7, a1));
        WK_ROW_P1(4, (void)0, (void)0, (void)0, (void)0);
        WK_ROW_P1(5, (void)0, (void)0, (void)0, (void)0);
        WK_ROW_P1(6, (void)0, (void)0, (void)0, (void)0);
        WK_ROW_P1(7, (void)0, (void)0, (void)0, (void)0);
#undef WK_ROW_P1
        WK_LGKM0(1);
        __builtin_amdgcn_s_barrier();                                            // every wave is done with stage it & 1
        WK_PIN();
        // ---- P2: half 1 from registers; iteration it + 2 goes out into the freed stage, then iteration it + 1 is awaited and its half 0 read
#define WK_DMA(Q) do { if (more2) issue_piece(Q, it & 1, it2); WK_PIN(); } while (0)
        // (DMA issue is spread thin -- one piece per four MFMAs: the four waves issue in step, and the texture-address unit takes 16 cycles
        //  per 1-KiB piece; a burst of pieces stalls the issuing wave, and its MFMAs with it)
#define WK_ROW_P2A(I, Q0)                                                                                            \
        WK_MF(1, I, 0); WK_DMA(Q0); WK_MF(1, I, 1); WK_MF(1, I, 2); WK_MF(1, I, 3);                                   \
        WK_MF(1, I, 4); WK_DMA(Q0 + 1); WK_MF(1, I, 5); WK_MF(1, I, 6); WK_MF(1, I, 7)
        WK_ROW_P2A(0, 0); WK_ROW_P2A(1, 2); WK_ROW_P2A(2, 4); WK_ROW_P2A(3, 6);
#undef WK_ROW_P2A
        if (more2) VMCNT(8); else VMCNT(0);                                      // this wave's pieces of iteration it + 1 (8 of it + 2 are younger) ...
        __builtin_amdgcn_s_barrier();                                            // ... and everyone's
        WK_PIN();
#define WK_RDN(X) do { if (more1) { X; } WK_PIN(); } while (0)
        // the 16 reads of the next half 0 sit in rows 4 - 6, so that the last of them has a row of MFMAs to land in before the next P1
        WK_MF(1, 4, 0); WK_RDN(WK_RDB(0, 0, b0n)); WK_MF(1, 4, 1); WK_RDN(WK_RDB(0, 1, b0n)); WK_MF(1, 4, 2); WK_DMA(8); WK_MF(1, 4, 3); WK_RDN(WK_RDB(0, 2, b0n));
        WK_MF(1, 4, 4); WK_RDN(WK_RDB(0, 3, b0n)); WK_MF(1, 4, 5); WK_RDN(WK_RDB(0, 4, b0n)); WK_MF(1, 4, 6); WK_DMA(9); WK_MF(1, 4, 7); WK_RDN(WK_RDB(0, 5, b0n));
        WK_MF(1, 5, 0); WK_RDN(WK_RDB(0, 6, b0n)); WK_MF(1, 5, 1); WK_RDN(WK_RDB(0, 7, b0n)); WK_MF(1, 5, 2); WK_DMA(10); WK_MF(1, 5, 3); WK_RDN(WK_RDA(0, 0, a0n));
        WK_MF(1, 5, 4); WK_RDN(WK_RDA(0, 1, a0n)); WK_MF(1, 5, 5); WK_RDN(WK_RDA(0, 2, a0n)); WK_MF(1, 5, 6); WK_DMA(11); WK_MF(1, 5, 7); WK_RDN(WK_RDA(0, 3, a0n));
        WK_MF(1, 6, 0); WK_RDN(WK_RDA(0, 4, a0n)); WK_MF(1, 6, 1); WK_RDN(WK_RDA(0, 5, a0n)); WK_MF(1, 6, 2); WK_DMA(12); WK_MF(1, 6, 3); WK_RDN(WK_RDA(0, 6, a0n));
        WK_MF(1, 6, 4); WK_RDN(WK_RDA(0, 7, a0n)); WK_MF(1, 6, 5); WK_MF(1, 6, 6); WK_DMA(13); WK_MF(1, 6, 7);
        WK_MF(1, 7, 0); WK_MF(1, 7, 1); WK_MF(1, 7, 2); WK_DMA(14); WK_MF(1, 7, 3); WK_MF(1, 7, 4); WK_MF(1, 7, 5); WK_MF(1, 7, 6); WK_DMA(15); WK_MF(1, 7, 7);
#undef WK_RDN
#define WK_ROW_P2B 0
#undef WK_ROW_P2B
#undef WK_DMA
        WK_LGKM0(0);
        {   // (a mask, not a branch: a branch here costs the loop its register assignment)
            const unsigned keep = (odd && it == nit - 2) ? 0u : 0xffffffffu;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint4 u = __builtin_bit_cast(uint4, af[0][i]);
                u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
                af[0][i] = __builtin_bit_cast(bf16x8, u);
            }
        }
        WK_PIN();
    }
    VMCNT(0);                                    // (the redundant pieces of the last two iterations: nothing may land in LDS after the workgroup has left)
#undef WK_RD
#undef WK_RDA
#undef WK_RDB
#undef WK_PIN
#undef WK_MF
#undef WK_LGKM0
    // the ring's epilogue per 64-column block: column half c of this wave is "wave column" wn * 2 + c of the 8-wave layout (same lane
    // ownership: 16 contiguous columns of one row), so every fused output of nt256_epilogue is available here too
    nt256_epilogue<EPI, F16, 4>(p, acc[0], m0, n0, wm, wn * 2, lane, oC);
    nt256_epilogue<EPI, F16, 4>(p, acc[1], m0, n0, wm, wn * 2 + 1, lane, oC);
}

// ---------------------------------------------------------------------------------------------
// TN :  P[z][n1][n2] = sum over token rows m of split z of A[m][n1] * B[m][n2]   (fp32 partials)
// ---------------------------------------------------------------------------------------------
constexpr int TK = 32;                       // token rows per step
constexpr int TN_TILE_BYTES = TK * 128 * 2;  // 8 KiB

__device__ __forceinline__ int tn_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
// byte offset of 16-byte chunk cc (0..15) of token row `row` in a [32][128] bf16 tile (32-byte swizzle)
__device__ __forceinline__ int tn_lds_off(int row, int cc) { return row * 256 + ((((cc >> 1) ^ tn_f(row)) << 5) | ((cc & 1) << 4)); }
// byte offset of element column `col` (multiple of 4) of row `row`
__device__ __forceinline__ int tn_lds_elem(int row, int col) { return row * 256 + ((((col >> 4) ^ tn_f(row)) << 5) | ((col & 15) << 1)); }

__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int colbase, int lane) {
    // MFMA operand with the reduction (token) index along the per-lane vector: lane (c = lane&15,
    // g = lane>>4) needs tile[8g + j][colbase + c], j = 0..7 -> two transposing LDS reads
    // (ds_read_b64_tr_b16: within a 16-lane group, lane i receives element (i&3) of the 8-byte
    // chunks addressed by lanes 4j + (i>>2), j = 0..3).
    const int t = lane & 15, g = lane >> 4;
    const int row = 8 * g + (t >> 2), col = colbase + ((t & 3) << 2);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tn_lds_elem(row, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tn_lds_elem(row + 4, col)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool X3, bool SHIFT>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT_ = X3 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int N1 = p.M, N2 = p.N;            // output dims; reduction length = p.K token rows
    const int tmi = blockIdx.x / p.tiles_n, tni = blockIdx.x % p.tiles_n;
    const int a0 = tmi * 128, b0 = tni * 128;
    const long long bz = blockIdx.y;
    const int z = blockIdx.z;
    const long long oA = boff(p, bz, p.sA, p.sA_in), oB = boff(p, bz, p.sB, p.sB_in);
    const bf16_t* A = p.A + oA;
    const bf16_t* B = p.B + oB;
    const bf16_t* Alo = X3 ? p.Alo + oA : nullptr;
    const bf16_t* Blo = X3 ? p.Blo + oB : nullptr;
    const long long mbeg = (long long)z * p.ksplit_len;
    const long long mend = min((long long)p.K, mbeg + p.ksplit_len);

    const int lrow = tid >> 4, lcc = tid & 15;   // 2 chunks per thread per operand: rows lrow, lrow+16
    const int quarter = SHIFT ? (p.shift_dim >> 2) : 1;
    uint4 ra[2], rb[2], ral[2], rbl[2];
    auto load_tile = [&](long long mk0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long g = mk0 + lrow + 16 * i;
            const bool rin = g < mend;
            const int ca = a0 + lcc * 8, cb = b0 + lcc * 8;
            const bool oka = rin && ca < N1;
            ra[i] = oka ? ldg16(A + g * p.lda + ca) : make_uint4(0, 0, 0, 0);
            if (X3) ral[i] = oka ? ldg16(Alo + g * p.lda + ca) : make_uint4(0, 0, 0, 0);
            bool okb = rin && cb < N2;
            long long gb = g;
            if (SHIFT && okb) {
                const ShiftRow s = shift_row(g, p.shift_ntok, p.shift_fmap);
                const int off = shift_pick(s, cb, quarter);
                if (off == INT_MIN) okb = false; else gb += off;
            }
            rb[i] = okb ? ldg16(B + gb * p.ldb + cb) : make_uint4(0, 0, 0, 0);
            if (X3) rbl[i] = okb ? ldg16(Blo + gb * p.ldb + cb) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * NT_ * TN_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = tn_lds_off(lrow + 16 * i, lcc);
            *reinterpret_cast<uint4*>(base + off) = ra[i];
            *reinterpret_cast<uint4*>(base + TN_TILE_BYTES + off) = rb[i];
            if (X3) {
                *reinterpret_cast<uint4*>(base + 2 * TN_TILE_BYTES + off) = ral[i];
                *reinterpret_cast<uint4*>(base + 3 * TN_TILE_BYTES + off) = rbl[i];
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((mend - mbeg + TK - 1) / TK);
    if (nk > 0) {
        load_tile(mbeg);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(mbeg + (long long)(kt + 1) * TK);
        const char* base = smem + cur * NT_ * TN_TILE_BYTES;
        bf16x8 af[4], bfr[4], afl[4], bfl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = tn_frag(base, wm * 64 + i * 16, lane);
            bfr[i] = tn_frag(base + TN_TILE_BYTES, wn * 64 + i * 16, lane);
            if (X3) {
                afl[i] = tn_frag(base + 2 * TN_TILE_BYTES, wm * 64 + i * 16, lane);
                bfl[i] = tn_frag(base + 3 * TN_TILE_BYTES, wn * 64 + i * 16, lane);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (X3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[j], af[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], afl[i], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }
    // partial[bz][z][n1][n2] (dense ld = N2)
    const int fr = lane & 15, fg = lane >> 4;
    float* P = partial + ((size_t)bz * gridDim.z + z) * (size_t)N1 * N2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n1 = a0 + wm * 64 + i * 16 + fr;
        if (n1 >= N1) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = b0 + wn * 64 + j * 16 + fg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n2 + r < N2) P[(size_t)n1 * N2 + n2 + r] = acc[i][j][r];
        }
    }
}

// TN, direct-to-LDS variant (bf16 operands only): same [32 token rows][128 columns] tiles and the same
// 32-byte XOR swizzle as gemm_tn_kernel, but filled by global_load_lds_dwordx4 with the swizzle applied to
// the per-lane source column.  One 1-KiB DMA piece = 4 token rows of a tile.
// NARROW (N2 <= 64, e.g. the per-head dK / dV GEMMs of the cross attention: [JP x 64] outputs): the four waves stack along
// M (32 rows x 64 columns each) instead of forming a 2 x 2 grid whose right half would multiply zeros.
// NS > 2: NS-stage DMA ring with a counted s_waitcnt (the per-step wait is a fraction of the memory latency instead of all of it;
// these few-tile GEMMs are a chain of k-steps with little arithmetic per step).
template <bool SHIFT, bool NARROW = false, int NS = 2>
__global__ __launch_bounds__(256) void gemm_tn_glds_kernel(GemmArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    constexpr int MI = NARROW ? 2 : 4, MSPAN = NARROW ? 32 : 64;      // row fragments per wave, rows per wave
    const int N1 = p.M, N2 = p.N;
    int tile = blockIdx.x;
    long long bz = blockIdx.y;
    if (NARROW) {
        // 1-D grid over (batch, tile): the column slabs of one operand matrix (the M tiles of a batch element) share cache lines
        // whenever the row pitch is not a multiple of 128 bytes, so they are placed on the SAME XCD (linear id mod 8) and next
        // to each other in dispatch order: the second and third slab then find those lines in that XCD's L2
        const int tiles = p.tiles_m * p.tiles_n, L = blockIdx.x, nbat = gridDim.x / tiles;
        if ((nbat & 7) == 0) { const int k = L >> 3; bz = (long long)(k / tiles) * 8 + (L & 7); tile = k % tiles; }
        else { bz = L / tiles; tile = L % tiles; }
    }
    const int tmi = tile / p.tiles_n, tni = tile % p.tiles_n;
    const int a0 = tmi * 128, b0 = tni * 128;
    const int z = blockIdx.z;
    const bf16_t* A = p.A + boff(p, bz, p.sA, p.sA_in);
    const bf16_t* B = p.B + boff(p, bz, p.sB, p.sB_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);
    const long long mbeg = (long long)z * p.ksplit_len;
    const long long mend = min((long long)p.K, mbeg + p.ksplit_len);
    const int quarter = SHIFT ? (p.shift_dim >> 2) : 1;
    const bool pow2 = SHIFT && (p.shift_fmap & (p.shift_fmap - 1)) == 0;
    const int fsh = pow2 ? __ffs(p.shift_fmap) - 1 : 0;

    // this lane's (row-in-tile, source column) for its two DMA pieces per operand
    int prow[2], cola[2], colb[2];
    bool oka[2], okb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 4 + wave) * 4 + (lane >> 4);
        const int ps = lane & 15;
        const int cc16 = ((((ps >> 1) ^ tn_f(row)) << 1) | (ps & 1));
        prow[j] = row;
        cola[j] = a0 + cc16 * 8; colb[j] = b0 + cc16 * 8;
        oka[j] = cola[j] < N1; okb[j] = colb[j] < N2;
    }
    auto issue = [&](int buf, long long mk0) {
        char* base = smem + buf * 2 * TN_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long g = mk0 + prow[j];
            const bool rin = g < mend;
            const bf16_t* sa = (rin && oka[j]) ? A + g * p.lda + cola[j] : zp;
            const bf16_t* sb = zp;
            if (rin && okb[j]) {
                long long gb = g;
                bool zero = false;
                if (SHIFT) {
                    const int i = (int)((unsigned)g % (unsigned)p.shift_ntok);
                    if (i > 0) {
                        const int pp = i - 1;
                        const int w = pow2 ? (pp & (p.shift_fmap - 1)) : pp % p.shift_fmap;
                        const int y = pow2 ? ((pp >> fsh) & (p.shift_fmap - 1)) : (pp / p.shift_fmap) % p.shift_fmap;
                        const int q = colb[j] / quarter;
                        if (q == 0) { if (y > 0) gb -= p.shift_fmap; else zero = true; }
                        else if (q == 1) { if (w > 0) gb -= 1; else zero = true; }
                    }
                }
                if (!zero) sb = B + gb * p.ldb + colb[j];
            }
            const int off = (j * 4 + wave) * 1024;
            // NARROW: columns beyond the operand are never read back (A rows >= N1 are not stored, B columns >= 64 not
            // multiplied), so those lanes skip their DMA piece instead of fetching the zero page (some lanes of every piece are
            // always active: the instruction count that s_waitcnt tracks does not change)
            // asm pieces (common.h): every wait on them below is explicit
            if (!NARROW || oka[j]) dma16_asm(sa, base + off);
            if (!NARROW || okb[j]) dma16_asm(sb, base + TN_TILE_BYTES + off);
        }
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((mend - mbeg + TK - 1) / TK);
    if (NS == 2) {
        if (nk > 0) issue(0, mbeg);
        VMCNT(0);
        __syncthreads();
    } else {
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < nk) issue(st, mbeg + (long long)st * TK);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = NS == 2 ? (kt & 1) : kt % NS;
        if (NS == 2) {
            if (kt + 1 < nk) issue(cur ^ 1, mbeg + (long long)(kt + 1) * TK);
        } else {
            // stage kt has landed once at most the later stages' loads (4 per stage and lane) are outstanding; the barrier also
            // tells every wave that stage kt - 1 is consumed, so its buffer can take stage kt + NS - 1
            if (kt + NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, mbeg + (long long)(kt + NS - 1) * TK);
        }
        const char* base = smem + cur * 2 * TN_TILE_BYTES;
        bf16x8 af[MI], bfr[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = tn_frag(base, wm * MSPAN + i * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) bfr[i] = tn_frag(base + TN_TILE_BYTES, wn * 64 + i * 16, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if (NS == 2) { VMCNT(0); __syncthreads(); }      // the other buffer has landed (it travelled under the MFMAs above); this one is free
    }
    const int fr = lane & 15, fg = lane >> 4;
    float* P = partial + ((size_t)bz * gridDim.z + z) * (size_t)N1 * N2;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int n1 = a0 + wm * MSPAN + i * 16 + fr;
        if (n1 >= N1) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = b0 + wn * 64 + j * 16 + fg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n2 + r < N2) P[(size_t)n1 * N2 + n2 + r] = acc[i][j][r];
        }
    }
}

// TN, WHOLE-M narrow form (round 5): the batched dK / dV products of the cross attention ([264 x 64] outputs, 2560 token rows, b * h
// batch elements).  On 128-row tiles a batch element was three workgroups -- the third with 8 live rows of 128 -- each walking all 80
// K-steps of 8 MFMAs per wave behind a barrier and each reading the whole B operand: a chain of latencies, not a stream.  Here ONE
// workgroup owns a batch element: MT [32][128] A tiles + one B tile per stage, NS-stage ring, the 16-row fragments dealt round-robin to
// the four waves (frag f -> wave f & 3), B read once.  Same fragment reads, same per-element order of accumulation (token rows ascending
// in steps of 32) as gemm_tn_glds_kernel: bit-identical results.  Every tile t < MT holds a live column (host: (MT - 1) * 128 < M), so
// every DMA instruction has an active lane and the vmcnt arithmetic is exact.  b = 128: 391 us against 425 + a 20-us reduction (five stages: 423).
template <int MT, int NS, bool F16 = false>     // F16: both operands hold fp16 values (the fp16-gradient backward of the cross attention)
__global__ __launch_bounds__(256) void gemm_tn_wm_kernel(GemmArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STG = (MT + 1) * TN_TILE_BYTES;
    constexpr int MI = 2 * MT;                                               // row fragments per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N1 = p.M, N2 = p.N;
    const long long bz = blockIdx.x;
    const int z = blockIdx.y;
    const bf16_t* A = p.A + boff(p, bz, p.sA, p.sA_in);
    const bf16_t* B = p.B + boff(p, bz, p.sB, p.sB_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);
    const long long mbeg = (long long)z * p.ksplit_len;
    const long long mend = min((long long)p.K, mbeg + p.ksplit_len);

    int prow[2], col[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 4 + wave) * 4 + (lane >> 4);
        const int ps = lane & 15;
        prow[j] = row;
        col[j] = ((((ps >> 1) ^ tn_f(row)) << 1) | (ps & 1)) * 8;
    }
    auto issue = [&](int buf, long long mk0) {
        char* base = smem + buf * STG;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long g = mk0 + prow[j];
            const bool rin = g < mend;
            const int off = (j * 4 + wave) * 1024;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int c = t * 128 + col[j];
                // (a_chunk: column c lives in plane c / 32 -- the cross-attention backward writes its dS / P' chunk by chunk, 16 queries x 64 B
                //  = 1 KiB contiguous per store instruction instead of 64-byte pieces at the row pitch; here 4 rows x 64 B per plane and piece)
                const bf16_t* src = p.a_chunk ? A + ((size_t)(c >> 5) * p.K + g) * 32 + (c & 31) : A + g * p.lda + c;
                if (c < N1) dma16_asm(rin ? src : zp, base + t * TN_TILE_BYTES + off);
            }
            if (col[j] < N2) dma16_asm(rin ? B + g * p.ldb + col[j] : zp, base + MT * TN_TILE_BYTES + off);
        }
    };
    constexpr int PER_STAGE = 2 * (MT + 1);                                    // vector-memory instructions per stage and wave

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((mend - mbeg + TK - 1) / TK);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) issue(st, mbeg + (long long)st * TK);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt % NS;
        if (kt + NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, mbeg + (long long)(kt + NS - 1) * TK);
        const char* base = smem + cur * STG;
        bf16x8 bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tn_frag(base + MT * TN_TILE_BYTES, j * 16, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int f = i * 4 + wave;
            if (f * 16 >= N1) continue;                                        // (wave-uniform)
            const bf16x8 af = tn_frag(base + (f >> 3) * TN_TILE_BYTES, (f & 7) * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(bfr[j], af, acc[i][j]);
        }
    }
    const int fr = lane & 15, fg = lane >> 4;
    // one split, beta = 0 (host: p.nsplit = 0): alpha * sum goes straight to C -- what splitk_reduce_kernel would have
    // written from the one partial (0 + x = x), without the round trip
    const bool direct = p.nsplit == 0;
    float* P = direct ? reinterpret_cast<float*>(p.C) + boff(p, bz, p.sC, p.sC_in) : partial + ((size_t)bz * gridDim.y + z) * (size_t)N1 * N2;
    const int ldp = direct ? p.ldc : N2;
    const float al = direct ? p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.f) : 1.f;
    const bool v4 = (N2 & 3) == 0 && (ldp & 3) == 0 && (reinterpret_cast<size_t>(P) & 15) == 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int n1 = (i * 4 + wave) * 16 + fr;
        if (n1 >= N1) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = j * 16 + fg * 4;
            if (direct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = al * acc[i][j][r];
            }
            if (v4 && n2 + 3 < N2) *reinterpret_cast<float4*>(P + (size_t)n1 * ldp + n2) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n2 + r < N2) P[(size_t)n1 * ldp + n2 + r] = acc[i][j][r];
        }
    }
}

// The same kernel with its per-K-step instruction count cut (round 6).  The counters of the form above (profiles/r06z_bench_b128_pmc_all.txt) show
// a kernel that is bound by instruction ISSUE, not by memory: 14 non-matrix vector instructions per MFMA (~235 per wave and K-step: 64-bit source
// addresses per DMA piece, per-lane live / tail selects, fragment addresses recomputed per read), waves active 58 % and waiting 13 % of their cycles.
// Here a DMA piece is ONE instruction: scalar base (advanced by a scalar add per K-step) + a per-lane 32-bit offset computed once; lanes of dead
// columns are clamped onto the last live 16-byte piece of their row (same cache line as a live lane: no extra traffic, no exec games, and the
// counted vmcnt stays exact); fragment addresses are six per-lane registers + immediates.  Host contract: whole K-steps only (rows of the split a
// multiple of 32) and plane offsets below 4 GiB -- everything else takes gemm_tn_wm_kernel.  Same loads into the same LDS places, same fragment
// reads, same MFMA order: bit-identical results.
template <int MT, int MI, bool F16>     // MI: row fragments per wave, ceil(ceil(M / 16) / 4) -- 2 MT - 1 or 2 MT; NO branch around an MFMA (see below)
__global__ __launch_bounds__(256) void gemm_tn_wmf_kernel(GemmArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 4;
    constexpr int STG = (MT + 1) * TN_TILE_BYTES;
    constexpr int PER_STAGE = 2 * (MT + 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N1 = p.M, N2 = p.N;
    const long long bz = blockIdx.x;
    const int z = blockIdx.y;
    const long long mbeg = (long long)z * p.ksplit_len;
    const long long mend = min((long long)p.K, mbeg + p.ksplit_len);
    const int nk = (int)((mend - mbeg) / TK);
    const unsigned rowA = p.a_chunk ? 64u : (unsigned)p.lda * 2u, rowB = (unsigned)p.ldb * 2u;           // bytes per token row
    const char* baseA = reinterpret_cast<const char*>(p.A + boff(p, bz, p.sA, p.sA_in)) + mbeg * rowA;
    const char* baseB = reinterpret_cast<const char*>(p.B + boff(p, bz, p.sB, p.sB_in)) + mbeg * rowB;
    const unsigned lds0 = (unsigned)(size_t)(lds_vptr_t)smem;

    unsigned offa[2][MT], offb[2];
    {
        const int cmaxA = (N1 - 1) & ~7, cmaxB = (N2 - 1) & ~7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (j * 4 + wave) * 4 + (lane >> 4);
            const int ps = lane & 15;
            const int col = ((((ps >> 1) ^ tn_f(row)) << 1) | (ps & 1)) * 8;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int c = min(t * 128 + col, cmaxA);
                offa[j][t] = p.a_chunk ? (unsigned)(c >> 5) * (unsigned)p.K * 64u + (unsigned)row * 64u + (unsigned)(c & 31) * 2u
                                       : (unsigned)row * rowA + (unsigned)c * 2u;
            }
            offb[j] = (unsigned)row * rowB + (unsigned)min(col, cmaxB) * 2u;
        }
    }
    auto piece = [&](unsigned voff, const char* sb, unsigned lds_addr) {
        const unsigned long long sbu = (unsigned long long)sb;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sbu), hi = __builtin_amdgcn_readfirstlane((unsigned)(sbu >> 32));
        const unsigned long long sbase = ((unsigned long long)hi << 32) | lo;
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
    };
    auto issue = [&](int buf, int kt) {
        const char* sa = baseA + (size_t)kt * (TK * rowA);
        const char* sb = baseB + (size_t)kt * (TK * rowB);
        const unsigned l = lds0 + buf * STG + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int t = 0; t < MT; ++t) piece(offa[j][t], sa, l + t * TN_TILE_BYTES + j * 4096);
            piece(offb[j], sb, l + MT * TN_TILE_BYTES + j * 4096);
        }
    };

    // fragment addresses (tn_frag): row 8 g + (t >> 2) (+ 4 for the second read: + 1024 bytes, same swizzle), 16-column group x -> (x ^ f) << 5
    unsigned fa[2], fb[4];
    {
        const int t = lane & 15, g = lane >> 4;
        const int row = 8 * g + (t >> 2), f = tn_f(row);
        const unsigned rb = lds0 + (unsigned)row * 256u + (unsigned)((t & 3) << 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) fa[h] = rb + (unsigned)(((h * 4 + wave) ^ f) << 5);     // this wave's row fragments: f = 4 i + wave -> group (i & 1) * 4 + wave of tile i >> 1
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = rb + (unsigned)((j ^ f) << 5) + MT * TN_TILE_BYTES;
    }
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto frag = [&](unsigned addr) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)addr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(addr + 1024u));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // Every wave multiplies all MI of its fragments, also the (at most three) waves whose last one lies past M (their products are not stored): a
    // wave-uniform branch around four MFMAs makes the compiler carry the accumulators through the loop in AGPRs and copy all of them to VGPRs and
    // back in EVERY iteration (192 v_accvgpr moves per 24 MFMAs in gemm_tn_wm_kernel: the issue bound named above).
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) issue(st, st);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue((kt + NS - 1) & (NS - 1), kt + NS - 1);
        const unsigned so = (unsigned)((kt & (NS - 1)) * STG);
        bf16x8 bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag(fb[j] + so);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const bf16x8 af = frag(fa[i & 1] + so + (i >> 1) * TN_TILE_BYTES);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(bfr[j], af, acc[i][j]);
        }
    }
    const int fr = lane & 15, fg = lane >> 4;
    const bool direct = p.nsplit == 0;                                       // (see gemm_tn_wm_kernel)
    float* P = direct ? reinterpret_cast<float*>(p.C) + boff(p, bz, p.sC, p.sC_in) : partial + ((size_t)bz * gridDim.y + z) * (size_t)N1 * N2;
    const int ldp = direct ? p.ldc : N2;
    const float al = direct ? p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.f) : 1.f;
    const bool v4 = (N2 & 3) == 0 && (ldp & 3) == 0 && (reinterpret_cast<size_t>(P) & 15) == 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int n1 = (i * 4 + wave) * 16 + fr;
        if (n1 >= N1) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = j * 16 + fg * 4;
            if (direct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = al * acc[i][j][r];
            }
            if (v4 && n2 + 3 < N2) *reinterpret_cast<float4*>(P + (size_t)n1 * ldp + n2) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n2 + r < N2) P[(size_t)n1 * ldp + n2 + r] = acc[i][j][r];
        }
    }
}

// TN, 256x256 output tile, 8 waves (2 x 4, 128 x 64 each), 32 token rows per K-step, NS-stage DMA ring with
// counted vmcnt + raw barrier (see gemm_nt_256_kernel).  Operand tiles are [32 rows][256 columns] (512-byte
// rows, 32-byte XOR swizzle on the source column), fragments come from ds_read_b64_tr_b16.
__device__ __forceinline__ int tn256_elem(int row, int col) { return row * 512 + ((((col >> 4) ^ tn_f(row)) << 5) | ((col & 15) << 1)); }
__device__ __forceinline__ bf16x8 tn256_frag(const char* tile, int colbase, int lane) {
    const int t = lane & 15, g = lane >> 4;
    const int row = 8 * g + (t >> 2), col = colbase + ((t & 3) << 2);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tn256_elem(row, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tn256_elem(row + 4, col)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool SHIFT, int NS, bool STAG = false>
__global__ __launch_bounds__(512) void gemm_tn_256_kernel(GemmArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 32 * 256 * 2;             // 16 KiB per operand per stage
    constexpr int STG = 2 * TB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int N1 = p.M, N2 = p.N;
    // flattened (split, tile) index, remapped so that the workgroups of one XCD are consecutive: the tiles of one K split
    // (which re-read the same token rows of A and B) then share that XCD's L2 instead of fetching the slice eight times
    const int ntile = p.tiles_m * p.tiles_n;
    const int vid = xcd_remap(blockIdx.x, ntile * p.nsplit);
    const int z = vid / ntile, tix = vid % ntile;
    const int tmi = tix / p.tiles_n, tni = tix % p.tiles_n;
    const int a0 = tmi * 256, b0 = tni * 256;
    const long long bz = blockIdx.y;
    const bf16_t* A = p.A + boff(p, bz, p.sA, p.sA_in);
    const bf16_t* B = p.B + boff(p, bz, p.sB, p.sB_in);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);
    const long long mbeg = (long long)z * p.ksplit_len;
    const long long mend = min((long long)p.K, mbeg + p.ksplit_len);
    const int quarter = SHIFT ? (p.shift_dim >> 2) : 1;
    const bool pow2 = SHIFT && (p.shift_fmap & (p.shift_fmap - 1)) == 0;
    const int fsh = pow2 ? __ffs(p.shift_fmap) - 1 : 0;

    // 16 DMA pieces (1 KiB = 2 token rows) per operand tile; each wave moves pieces wave and wave + 8
    int prow[2], cola[2], colb[2];
    bool oka[2], okb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 2 + (lane >> 5);
        const int ps = lane & 31;                               // 16-byte chunk position inside the 512-byte row
        const int cc16 = ((((ps >> 1) ^ tn_f(row)) << 1) | (ps & 1));
        prow[j] = row;
        cola[j] = a0 + cc16 * 8; colb[j] = b0 + cc16 * 8;
        oka[j] = cola[j] < N1; okb[j] = colb[j] < N2;
    }
    // Tiles are issued strictly in order, so every piece keeps running source pointers / row / token position and advances them
    // by TK rows per issue: no 64-bit multiply or modulo on the per-K-step path.
    int ipos[2] = {0, 0};
    int qb[2] = {2, 2};
    long long grow[2];
    const bf16_t* pA[2]; const bf16_t* pB[2];
    const long long stepA = (long long)TK * p.lda, stepB = (long long)TK * p.ldb;
    const long long dH = -(long long)p.shift_fmap * p.ldb, dW = -(long long)p.ldb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        grow[j] = mbeg + prow[j];
        pA[j] = A + grow[j] * p.lda + cola[j];
        pB[j] = B + grow[j] * p.ldb + colb[j];
        if (SHIFT) {
            ipos[j] = (int)((unsigned long long)grow[j] % (unsigned)p.shift_ntok);
            qb[j] = (colb[j] >= quarter) + (colb[j] >= 2 * quarter);
        }
    }
    auto issue = [&](int slot, long long) {
        char* base = smem + slot * STG;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool rin = grow[j] < mend;
            const bf16_t* sa = (rin && oka[j]) ? pA[j] : zp;
            const bf16_t* sb = (rin && okb[j]) ? pB[j] : zp;
            if (SHIFT) {
                const int i = ipos[j];
                if (i > 0 && qb[j] < 2 && rin && okb[j]) {
                    const int pp = i - 1;
                    if (qb[j] == 0) {
                        const int y = pow2 ? ((pp >> fsh) & (p.shift_fmap - 1)) : (pp / p.shift_fmap) % p.shift_fmap;
                        sb = y > 0 ? pB[j] + dH : zp;
                    } else {
                        const int w = pow2 ? (pp & (p.shift_fmap - 1)) : pp % p.shift_fmap;
                        sb = w > 0 ? pB[j] + dW : zp;
                    }
                }
                ipos[j] += TK;
                while (ipos[j] >= p.shift_ntok) ipos[j] -= p.shift_ntok;
            }
            grow[j] += TK; pA[j] += stepA; pB[j] += stepB;
            const int off = (j * 8 + wave) * 1024;
            dma16_asm(sa, base + off);               // (asm: see common.h -- the builtin makes the compiler drain the ring before every fragment read)
            dma16_asm(sb, base + TB + off);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((mend - mbeg + TK - 1) / TK);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, mbeg + (long long)s * TK);
    if constexpr (STAG) {                        // staggered wave rows: see gemm_nt_256_kernel
        static_assert(NS == 4, "staggered schedule is written for the 4-stage ring");
        if (nk >= 3) VMCNT(8); else if (nk == 2) VMCNT(4); else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const char* base = smem + (kt % NS) * STG;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tn256_frag(base + TB, wn * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = tn256_frag(base, wm * 128 + i * 16, lane);
            if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, mbeg + (long long)(kt + NS - 1) * TK);
            const int rem2 = nk - 2 - kt;
            if (rem2 >= 2) VMCNT(8); else if (rem2 == 1) VMCNT(4); else VMCNT(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        if (NS >= 4 && rem >= 2) VMCNT(8);
        else if (rem >= 1) VMCNT(4);
        else VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, mbeg + (long long)(kt + NS - 1) * TK);
        const char* base = smem + (kt % NS) * STG;
        bf16x8 af[8], bfr[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = tn256_frag(base, wm * 128 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tn256_frag(base + TB, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    const int fr = lane & 15, fg = lane >> 4;
    float* P = partial + ((size_t)bz * p.nsplit + z) * (size_t)N1 * N2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n1 = a0 + wm * 128 + i * 16 + fr;
        if (n1 >= N1) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = b0 + wn * 64 + j * 16 + fg * 4;
            if (n2 + 3 < N2 && (N2 & 3) == 0) *reinterpret_cast<float4*>(P + (size_t)n1 * N2 + n2) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n2 + r < N2) P[(size_t)n1 * N2 + n2 + r] = acc[i][j][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TN, 256x256 output tile, FOUR waves (2 x 2, 128x128 each), 64 token rows per iteration, two 64 KiB stages with a 1.5-iteration
// prefetch: the weight-gradient form of gemm_nt_w4k_kernel (same schedule, same reasons; see there).  Operand tiles are
// [64 token rows][256 columns] in the layout of gemm_tn_256_kernel (512-byte rows, 32-byte XOR swizzle), fragments come from
// ds_read_b64_tr_b16 pairs (two per fragment: 32 + 32 reads per K half).  No token shift; every split covers a multiple of 64 rows
// (host side); columns past the leading dimension are clamped (columns past N1 / N2 only feed outputs that are never stored).
// Accumulation order per output element = token rows ascending in chunks of 32, as gemm_tn_256_kernel: bit-identical partials.
// ---------------------------------------------------------------------------------------------
// F16: both operands hold fp16 values (fp16 gradients x the fp16 activation copies of the 'bf16x3-fwd' mode, round 5): the fp16 MFMA, nothing else differs.
template <bool F16>
__global__ __launch_bounds__(256, 1) void gemm_tn_w4k_kernel(GemmArgs p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * 256 * 2;             // one operand image of a stage: 32 KiB
    constexpr int STG = 2 * TB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int N1 = p.M, N2 = p.N;
    const int ntile = p.tiles_m * p.tiles_n;
    const int vid = xcd_remap(blockIdx.x, ntile * p.nsplit);
    const int z = vid / ntile, tix = vid % ntile;
    const int tmi = tix / p.tiles_n, tni = tix % p.tiles_n;
    const int a0 = tmi * 256, b0 = tni * 256;
    const long long bz = blockIdx.y;
    const long long mbeg = (long long)z * p.ksplit_len;
    const long long mend = min((long long)p.K, mbeg + p.ksplit_len);
    const int nit = mend > mbeg ? (int)((mend - mbeg) / 64) : 0;
    const unsigned lds0 = (unsigned)(size_t)(lds_vptr_t)smem;
    const char* baseA = reinterpret_cast<const char*>(p.A + boff(p, bz, p.sA, p.sA_in) + mbeg * p.lda);
    const char* baseB = reinterpret_cast<const char*>(p.B + boff(p, bz, p.sB, p.sB_in) + mbeg * p.ldb);
    // this wave's DMA pieces: pieces j * 4 + wave (j = 0..7) of each image; a piece = 2 token rows x 256 columns (1 KiB)
    unsigned offa[8], offb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j * 4 + wave) * 2 + (lane >> 5);
        const int ps = lane & 31;
        const int cc16 = ((((ps >> 1) ^ tn_f(row)) << 1) | (ps & 1));
        offa[j] = (unsigned)((row * p.lda + min(a0 + cc16 * 8, p.lda - 8)) * 2);         // (the row is readable up to its leading dimension)
        offb[j] = (unsigned)((row * p.ldb + min(b0 + cc16 * 8, p.ldb - 8)) * 2);
    }
    const size_t itA = (size_t)64 * p.lda * 2, itB = (size_t)64 * p.ldb * 2;
    auto issue_piece = [&](int q, int stage, int it) {            // q = 0..7: A pieces, 8..15: B pieces
        const char* sb = q < 8 ? baseA + it * itA : baseB + it * itB;
        const unsigned long long sbu = (unsigned long long)sb;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sbu), hi = __builtin_amdgcn_readfirstlane((unsigned)(sbu >> 32));
        const unsigned long long sbase = ((unsigned long long)hi << 32) | lo;
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane(lds0 + stage * STG + (q < 8 ? 0 : TB) + ((q & 7) * 4 + wave) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(q < 8 ? offa[q & 7] : offb[q & 7]), "s"(sbase), "s"(lds_addr)
                     : "memory", "m0");
    };
    f32x4 acc[2][8][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses in stage 0, K half 0 (half 1: + 16384; the `hi` read of a fragment: + 2048): the column block index is XORed with a
    // per-lane swizzle term, so every fragment keeps its own address register
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int t16 = lane & 15, g4 = lane >> 4;
    const int frow = 8 * g4 + (t16 >> 2), ff = tn_f(frow);
    unsigned ada[8], adb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ada[i] = lds0 + (unsigned)(frow * 512 + ((((wm * 8 + i) ^ ff) << 5) | ((t16 & 3) << 3)));
        adb[i] = lds0 + (unsigned)(TB + frow * 512 + ((((wn * 8 + i) ^ ff) << 5) | ((t16 & 3) << 3)));
    }
    bf16x8 af[2][8], bfr[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) af[h][i] = bfr[h][i] = bf16x8{};
    auto rd = [&](unsigned addr) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)addr);              // (a 32-bit LDS address, lds0 included: no generic-pointer arithmetic per read)
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(addr + 2048u));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
#define TK_PIN() __builtin_amdgcn_sched_barrier(0)
    // the fragments of a K half, named as VGPR operands where the half starts to be multiplied: without it the register allocator lets some
    // ds_read results land in AGPRs (legal on gfx950) and shuffles accumulator tuples around them, ~80 v_accvgpr moves per iteration
#define TK_VGPR(H)                                                                                                              \
    asm volatile("" : "+v"(af[H][0]), "+v"(af[H][1]), "+v"(af[H][2]), "+v"(af[H][3]), "+v"(af[H][4]), "+v"(af[H][5]), "+v"(af[H][6]),      \
                 "+v"(af[H][7]), "+v"(bfr[H][0]), "+v"(bfr[H][1]), "+v"(bfr[H][2]), "+v"(bfr[H][3]), "+v"(bfr[H][4]), "+v"(bfr[H][5]),   \
                 "+v"(bfr[H][6]), "+v"(bfr[H][7]))
#define TK_RDA(H, I, SO) do { af[H][I] = rd(ada[I] + (H) * 16384); TK_PIN(); } while (0)
#define TK_RDB(H, Q, SO) do { bfr[H][Q] = rd(adb[Q] + (H) * 16384); TK_PIN(); } while (0)
#define TK_MF(H, I, Q) do { acc[(Q) >> 2][I][(Q) & 3] = mfma16<F16>(bfr[H][Q], af[H][I], acc[(Q) >> 2][I][(Q) & 3]); TK_PIN(); } while (0)
    if (nit > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) issue_piece(q, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) issue_piece(q, 1, min(1, nit - 1));
        VMCNT(16);
        __builtin_amdgcn_s_barrier();
        TK_RDB(0, 0, 0u); TK_RDB(0, 1, 0u); TK_RDB(0, 2, 0u); TK_RDB(0, 3, 0u); TK_RDB(0, 4, 0u); TK_RDB(0, 5, 0u); TK_RDB(0, 6, 0u); TK_RDB(0, 7, 0u);
        TK_RDA(0, 0, 0u); TK_RDA(0, 1, 0u); TK_RDA(0, 2, 0u); TK_RDA(0, 3, 0u); TK_RDA(0, 4, 0u); TK_RDA(0, 5, 0u); TK_RDA(0, 6, 0u); TK_RDA(0, 7, 0u);
    }
    TK_PIN();
    // (one branch-free body: past the end of the split the staging repeats the last step into the stage just consumed -- see gemm_nt_w4k_kernel)
    for (int it = 0; it < nit; ++it) {
        const unsigned so = 0u, sn = 0u;             // (the 16 fragment address registers follow the stage: flipped once per iteration, below)
        const int flip = (it & 1) ? -STG : STG;
        const int it2 = min(it + 2, nit - 1);
        TK_VGPR(0);
        TK_PIN();
#define TK_DMA(Q) do { issue_piece(Q, it & 1, it2); TK_PIN(); } while (0)
        // ---- P1: half 0 from registers; the reads of half 1 (16 fragments = 32 reads) between the MFMAs of rows 0 - 3
#define TK_ROW_P1(I, R0, R1, R2, R3)                                                                                \
        TK_MF(0, I, 0); R0; TK_MF(0, I, 1); TK_MF(0, I, 2); R1; TK_MF(0, I, 3); TK_MF(0, I, 4); R2; TK_MF(0, I, 5); TK_MF(0, I, 6); R3; TK_MF(0, I, 7)
        TK_ROW_P1(0, TK_RDB(1, 0, so), TK_RDB(1, 1, so), TK_RDB(1, 2, so), TK_RDB(1, 3, so));
        TK_ROW_P1(1, TK_RDB(1, 4, so), TK_RDB(1, 5, so), TK_RDB(1, 6, so), TK_RDB(1, 7, so));
        TK_ROW_P1(2, TK_RDA(1, 0, so), TK_RDA(1, 1, so), TK_RDA(1, 2, so), TK_RDA(1, 3, so));
        TK_ROW_P1(3, TK_RDA(1, 4, so), TK_RDA(1, 5, so), TK_RDA(1, 6, so), TK_RDA(1, 7, so));
        TK_ROW_P1(4, (void)0, (void)0, (void)0, (void)0); TK_ROW_P1(5, (void)0, (void)0, (void)0, (void)0);
        TK_ROW_P1(6, (void)0, (void)0, (void)0, (void)0); TK_ROW_P1(7, (void)0, (void)0, (void)0, (void)0);
#undef TK_ROW_P1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                            // every wave is done with stage it & 1
        TK_VGPR(1);
        TK_PIN();
        // ---- P2: half 1 from registers; eight pieces of iteration it + 2, then iteration it + 1 is awaited and its half 0 read (rows 4 - 6)
#define TK_FLIP(X) do { ada[X] += flip; adb[X] += flip; TK_PIN(); } while (0)
#define TK_ROW_P2A(I, Q0)                                                                                            \
        TK_MF(1, I, 0); TK_DMA(Q0); TK_MF(1, I, 1); TK_MF(1, I, 2); TK_FLIP(2 * ((I) & 3)); TK_MF(1, I, 3); TK_MF(1, I, 4); TK_DMA(Q0 + 1);  \
        TK_MF(1, I, 5); TK_MF(1, I, 6); TK_FLIP(2 * ((I) & 3) + 1); TK_MF(1, I, 7)
        TK_ROW_P2A(0, 0); TK_ROW_P2A(1, 2); TK_ROW_P2A(2, 4); TK_ROW_P2A(3, 6);
#undef TK_FLIP
#undef TK_ROW_P2A
        VMCNT(8);
        __builtin_amdgcn_s_barrier();
        TK_PIN();
        TK_MF(1, 4, 0); TK_RDB(0, 0, sn); TK_MF(1, 4, 1); TK_RDB(0, 1, sn); TK_MF(1, 4, 2); TK_DMA(8); TK_MF(1, 4, 3); TK_RDB(0, 2, sn);
        TK_MF(1, 4, 4); TK_RDB(0, 3, sn); TK_MF(1, 4, 5); TK_RDB(0, 4, sn); TK_MF(1, 4, 6); TK_DMA(9); TK_MF(1, 4, 7); TK_RDB(0, 5, sn);
        TK_MF(1, 5, 0); TK_RDB(0, 6, sn); TK_MF(1, 5, 1); TK_RDB(0, 7, sn); TK_MF(1, 5, 2); TK_DMA(10); TK_MF(1, 5, 3); TK_RDA(0, 0, sn);
        TK_MF(1, 5, 4); TK_RDA(0, 1, sn); TK_MF(1, 5, 5); TK_RDA(0, 2, sn); TK_MF(1, 5, 6); TK_DMA(11); TK_MF(1, 5, 7); TK_RDA(0, 3, sn);
        TK_MF(1, 6, 0); TK_RDA(0, 4, sn); TK_MF(1, 6, 1); TK_RDA(0, 5, sn); TK_MF(1, 6, 2); TK_DMA(12); TK_MF(1, 6, 3); TK_RDA(0, 6, sn);
        TK_MF(1, 6, 4); TK_RDA(0, 7, sn); TK_MF(1, 6, 5); TK_MF(1, 6, 6); TK_DMA(13); TK_MF(1, 6, 7);
        TK_MF(1, 7, 0); TK_MF(1, 7, 1); TK_MF(1, 7, 2); TK_DMA(14); TK_MF(1, 7, 3); TK_MF(1, 7, 4); TK_MF(1, 7, 5); TK_MF(1, 7, 6); TK_DMA(15); TK_MF(1, 7, 7);
#undef TK_DMA
    }
    VMCNT(0);
#undef TK_PIN
#undef TK_VGPR
#undef TK_RDA
#undef TK_RDB
#undef TK_MF
    const int fr = lane & 15, fg = lane >> 4;
    float* P = partial + ((size_t)bz * p.nsplit + z) * (size_t)N1 * N2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n1 = a0 + wm * 128 + i * 16 + fr;
        if (n1 >= N1) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n2 = b0 + wn * 128 + q * 16 + fg * 4;
            const f32x4 v = acc[q >> 2][i][q & 3];
            if (n2 + 3 < N2 && (N2 & 3) == 0) *reinterpret_cast<float4*>(P + (size_t)n1 * N2 + n2) = make_float4(v[0], v[1], v[2], v[3]);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n2 + r < N2) P[(size_t)n1 * N2 + n2 + r] = v[r];
        }
    }
}

// C[bz][n1][n2] = beta*C + alpha * sum_z partial[bz][z][n1][n2]   (fixed order -> deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C, long long sC, long long sC_in,
                                     int batch_inner, int ldc, int N1, int N2, int splits, float alpha, float beta,
                                     const float* __restrict__ alpha_dev) {
    if (alpha_dev) alpha *= *alpha_dev;          // (1 / S of an fp16-gradient backward: a device scalar, no host synchronisation)
    const size_t per = (size_t)N1 * N2;
    const int bz = blockIdx.y;
    if ((N2 & 3) == 0 && (ldc & 3) == 0 && (sC & 3) == 0 && (sC_in & 3) == 0 && (reinterpret_cast<size_t>(C) & 15) == 0) {
        // four columns per thread, eight splits in flight: the scalar form below issued ONE 4-byte load per iteration and waited for it
        // (dependent add).  The adds keep their order (split 0, 1, 2, ...), so the sums are bit-identical to the scalar form's.
        const size_t per4 = per >> 2;
        for (size_t e4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e4 < per4; e4 += (size_t)gridDim.x * blockDim.x) {
            const size_t e = e4 << 2;
            const float* P = partial + (size_t)bz * splits * per + e;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int z = 0;
            for (; z + 8 <= splits; z += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(P + (size_t)(z + u) * per);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            for (; z < splits; ++z) { const float4 v = *reinterpret_cast<const float4*>(P + (size_t)z * per); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            const int n1 = (int)(e / N2), n2 = (int)(e % N2);
            const long long oC = batch_inner > 0 ? (long long)(bz / batch_inner) * sC + (long long)(bz % batch_inner) * sC_in : (long long)bz * sC;
            float* c = C + oC + (size_t)n1 * ldc + n2;
            float4 o = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
            if (beta != 0.f) { const float4 cv = *reinterpret_cast<const float4*>(c); o.x = beta * cv.x + o.x; o.y = beta * cv.y + o.y; o.z = beta * cv.z + o.z; o.w = beta * cv.w + o.w; }
            *reinterpret_cast<float4*>(c) = o;
        }
        return;
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        const float* P = partial + (size_t)bz * splits * per + e;
        for (int z = 0; z < splits; ++z) s += P[(size_t)z * per];
        const int n1 = (int)(e / N2), n2 = (int)(e % N2);
        const long long oC = batch_inner > 0 ? (long long)(bz / batch_inner) * sC + (long long)(bz % batch_inner) * sC_in : (long long)bz * sC;
        float* c = C + oC + (size_t)n1 * ldc + n2;
        *c = (beta != 0.f ? beta * *c : 0.f) + alpha * s;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Few-row NT GEMM (M <= 8 per workgroup pass): the decode step of generate() multiplies ONE new row per sample by every
// weight matrix, so the operation is a stream over the weight (HBM / L2 bound) and the 128x128 / 256x256 MFMA tiles would
// spend a whole tile round on a handful of rows.  A[m, :] lives in LDS as fp32 (hi + lo); each wave owns ROWS_CPW output
// columns, every lane multiplies its 8-wide k-slice of those weight rows with the matching slice of all rows of A, and the
// partial sums are reduced across the wave.  fp32 FMA throughout (more exact than the bf16x3 MFMA path).
// ---------------------------------------------------------------------------------------------
constexpr int ROWS_MR = 8, ROWS_CPW = 4;

// one halving step of the wave reduction (compile-time indices only: the values stay in registers)
template <int HALF, int BIT>
__device__ __forceinline__ void rows_halve(float (&v)[ROWS_CPW * ROWS_MR], int lane) {
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float send = up ? v[i] : v[HALF + i];
        const float keep = up ? v[HALF + i] : v[i];
        v[i] = keep + __shfl_xor(send, BIT, 64);
    }
    if constexpr (HALF > 1) rows_halve<HALF / 2, BIT / 2>(v, lane);
}

template <bool LO>
__global__ __launch_bounds__(256) void gemm_nt_rows_kernel(GemmArgs p) {
    extern __shared__ float a_s[];                       // [ROWS_MR][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * ROWS_MR;
    const int K = p.K;
    const int n0 = (blockIdx.x * 4 + wave) * ROWS_CPW;
    // the first weight slices are requested before A is staged: the two memory latencies overlap
    uint4 wv[ROWS_CPW], wl[ROWS_CPW];
    if (lane * 8 < K) {
#pragma unroll
        for (int c = 0; c < ROWS_CPW; ++c) {
            const int n = min(n0 + c, p.N - 1);
            wv[c] = ldg16(p.B + (size_t)n * p.ldb + lane * 8);
            if (LO) wl[c] = ldg16(p.Blo + (size_t)n * p.ldb + lane * 8);
        }
    }
    for (int i = tid * 8; i < ROWS_MR * K; i += 256 * 8) {
        const int m = i / K, k = i - m * K;
        float v[8];
        if (m0 + m < p.M) {
            const uint4 h = ldg16(p.A + (size_t)(m0 + m) * p.lda + k);
            v[0] = lo_f(h.x); v[1] = hi_f(h.x); v[2] = lo_f(h.y); v[3] = hi_f(h.y);
            v[4] = lo_f(h.z); v[5] = hi_f(h.z); v[6] = lo_f(h.w); v[7] = hi_f(h.w);
            if (LO) {
                const uint4 l = ldg16(p.Alo + (size_t)(m0 + m) * p.lda + k);
                v[0] += lo_f(l.x); v[1] += hi_f(l.x); v[2] += lo_f(l.y); v[3] += hi_f(l.y);
                v[4] += lo_f(l.z); v[5] += hi_f(l.z); v[6] += lo_f(l.w); v[7] += hi_f(l.w);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        *reinterpret_cast<float4*>(a_s + i) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(a_s + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    if (n0 >= p.N) return;
    float acc[ROWS_CPW][ROWS_MR];
#pragma unroll
    for (int c = 0; c < ROWS_CPW; ++c)
#pragma unroll
        for (int m = 0; m < ROWS_MR; ++m) acc[c][m] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        if (k >= 512) {
#pragma unroll
            for (int c = 0; c < ROWS_CPW; ++c) {
                const int n = min(n0 + c, p.N - 1);
                wv[c] = ldg16(p.B + (size_t)n * p.ldb + k);
                if (LO) wl[c] = ldg16(p.Blo + (size_t)n * p.ldb + k);
            }
        }
        float w[ROWS_CPW][8];
#pragma unroll
        for (int c = 0; c < ROWS_CPW; ++c) {
            w[c][0] = lo_f(wv[c].x); w[c][1] = hi_f(wv[c].x); w[c][2] = lo_f(wv[c].y); w[c][3] = hi_f(wv[c].y);
            w[c][4] = lo_f(wv[c].z); w[c][5] = hi_f(wv[c].z); w[c][6] = lo_f(wv[c].w); w[c][7] = hi_f(wv[c].w);
            if (LO) {
                w[c][0] += lo_f(wl[c].x); w[c][1] += hi_f(wl[c].x); w[c][2] += lo_f(wl[c].y); w[c][3] += hi_f(wl[c].y);
                w[c][4] += lo_f(wl[c].z); w[c][5] += hi_f(wl[c].z); w[c][6] += lo_f(wl[c].w); w[c][7] += hi_f(wl[c].w);
            }
        }
#pragma unroll
        for (int m = 0; m < ROWS_MR; ++m) {
            const float4 a0 = *reinterpret_cast<const float4*>(a_s + m * K + k);
            const float4 a1 = *reinterpret_cast<const float4*>(a_s + m * K + k + 4);
#pragma unroll
            for (int c = 0; c < ROWS_CPW; ++c) {
                float t = acc[c][m];
                t = fmaf(a0.x, w[c][0], t); t = fmaf(a0.y, w[c][1], t); t = fmaf(a0.z, w[c][2], t); t = fmaf(a0.w, w[c][3], t);
                t = fmaf(a1.x, w[c][4], t); t = fmaf(a1.y, w[c][5], t); t = fmaf(a1.z, w[c][6], t); t = fmaf(a1.w, w[c][7], t);
                acc[c][m] = t;
            }
        }
    }
    // wave reduction of the 32 partial sums by halving: at each step a lane keeps half of its values and receives the partner's
    // copies of that half (32 cross-lane moves instead of 32 x 6); lane 2 * i (and 2 * i + 1) ends with the total of value i
    float v[ROWS_CPW * ROWS_MR];
#pragma unroll
    for (int c = 0; c < ROWS_CPW; ++c)
#pragma unroll
        for (int m = 0; m < ROWS_MR; ++m) v[c * ROWS_MR + m] = acc[c][m];
    rows_halve<ROWS_CPW * ROWS_MR / 2, 32>(v, lane);
    const float mine = v[0] + __shfl_xor(v[0], 1, 64);
    static_assert(ROWS_CPW * ROWS_MR == 32, "the halving reduction is written for 32 values over 64 lanes");
    if ((lane & 1) == 0) {
        const int vi = lane >> 1, c = vi / ROWS_MR, m = m0 + vi % ROWS_MR, n = n0 + c;
        if (m < p.M && n < p.N) {
            float v = mine * p.alpha;
            if (p.bias) v += p.bias[n];
            const size_t o = (size_t)m * p.ldc + n;
            if (p.Clo) { bf16_t h, l; f2bf_hilo(v, h, l); ((bf16_t*)p.C)[o] = h; p.Clo[o] = l; }
            else if (p.beta != 0.f) ((bf16_t*)p.C)[o] = f2bf(v);          // beta != 0 marks a bf16 destination here
            else ((float*)p.C)[o] = v;
        }
    }
}

extern "C" int amdnuwa_geglu_il_fwd(const uint16_t* u_hi, const uint16_t* u_lo, uint16_t* o_hi, uint16_t* o_lo, long long R, int FP, hipStream_t stream);

extern "C" int amdnuwa_geglu_il_bwd(const uint16_t* u_hi, const uint16_t* u_lo, const uint16_t* d_hi, const uint16_t* d_lo,
                                    uint16_t* du_hi, uint16_t* du_lo, long long R, int FP, hipStream_t stream);

// can the GEGLU gate ride in the epilogue of the kernel this product will run on (the 256x256 staggered ring, bf16 output)?
static bool nt_geglu_fusable(const amdnuwa_gemm_desc* d) {
    if (!d->c_is_bf16 || d->batch > 1) return false;
    if (d->Alo) {                  // bf16x3 ring: the FORWARD gate only (hi + lo gate output), no token shift in the loader
        if (!d->Blo || d->geglu_u || d->shift_ntok > 0 || g_amdnuwa_tuning[13] == 1) return false;
    } else if (d->Clo || d->C2lo) return false;
    if (d->K % 32 || d->N % 16 || d->ldc % 8 || d->ldc2 % 8) return false;
    if (d->geglu_u && (d->geglu_u_lo || d->ld_u % 8)) return false;
    const int v = g_amdnuwa_tuning[0];
    if (v == 7) return true;
    if (v != 0 || d->M <= 4 * ROWS_MR) return false;
    return (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) >= 512;
}

// start-phase step for the first-generation workgroups (see gemm_nt_256_kernel).  Measured (r02, b = 64): a perfectly regular grid (qkv:
// 3840 equal tiles = 15 per CU) stays phase-locked and gains 12 % in isolation (380 -> 335 us at step 12), ragged grids (ff1, dgrad ff2)
// gain nothing, and inside the training step -- where the previous kernel's tail already staggers the CUs -- the whole effect is
// below the noise (600.3 k vs 600.3 k tokens/s).  Off unless forced through tuning key 14 (> 0 = phase step in s_sleep(8) units).
static int nt_skew(long long tiles) {
    (void)tiles;
    const int t = g_amdnuwa_tuning[14];
    return t > 0 ? t : 0;
}

// ---- fused linear + cross entropy: row statistics -> lse, row loss, mean (fixed order) ------------------------------------------
namespace {
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* __restrict__ stats, const float* __restrict__ tl, int nblk,
                                                         long long R, float* __restrict__ lse, float* __restrict__ row_loss) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= R) return;
    const float2* st = reinterpret_cast<const float2*>(stats) + row * nblk;
    float m = st[0].x;
    for (int b = 1; b < nblk; ++b) m = fmaxf(m, st[b].x);
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += st[b].y * __expf(st[b].x - m);
    const float l = m + __logf(s);
    lse[row] = l;
    row_loss[row] = l - tl[row];                       // tl is NaN where the target id lies outside [0, C)
}
__global__ __launch_bounds__(1024) void ce_mean_kernel(const float* __restrict__ v, long long n, float* __restrict__ out) {
    __shared__ float sred[1024];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) s += v[i];
    sred[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sred[0] / (float)n;
}
}  // namespace

// persistent form of the staggered 256x256 ring (gemm_nt_256p_kernel): taken when a CU owns more than one tile and the main loop is short
// (K <= 1024); tuning key 20 = 1 keeps the one-tile-per-workgroup launch everywhere, = 2 forces the persistent kernel for any shape
static int nt_persistent_grid() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    return cus;
}
// the four-wave K-step 64 kernel (gemm_nt_w4k_kernel): the long-K products (tuning key 22 = 1 keeps the 8-wave ring, = 2 takes it for
// every K that is a multiple of 64)
static bool nt_long_k(int K, int dbg) {
    const int v = g_amdnuwa_tuning[22];
    if (v == 1 || K % 32 || K < 96 || (dbg & 2)) return false;
    // (K % 64 == 32 -- FF2's 1376 -- works, bit-identical, and ties the K-step 64 ring at 592-606 vs 575-662 us: only on request)
    return v == 2 || (K >= 1024 && K % 64 == 0);
}
static bool nt_persistent(long long tiles, int K, int dbg) {
    const int v = g_amdnuwa_tuning[20];
    if (v == 1 || K % 32 || K / 32 < 3 || (dbg & 2)) return false;
    // measured at b = 128 (tools/gemm_persist_probe.py): K = 512 shapes -2...-9 %, K = 1536 a tie, K >= 2752 +2...+5 % (the ring fill is
    // nothing next to a long main loop there, and the one-tile launch lets the hardware balance the tail)
    return v == 2 || (tiles > nt_persistent_grid() && K <= 1024);
}

extern "C" size_t amdnuwa_linear_ce_workspace_bytes(long long R, int C) {
    if (R <= 0 || C <= 0 || C % 64) return 0;
    return (size_t)R * (C / 64) * 2 * sizeof(float) + (size_t)R * 2 * sizeof(float);
}

// h_lo / w_lo != NULL: the statistics pass on the hi + lo ring (three MFMAs per product: the logits of the 'bf16x3' modes); its
// dlogits pass too, unless fp16 copies of both operands are given (h16 / w16): then dlogits -- a bf16 tensor for a bf16 backward --
// come from ONE fp16 MFMA per product against the exact lse of pass 1
static int linear_ce_run(const uint16_t* h, const uint16_t* h_lo, const uint16_t* h16, int ldh, const uint16_t* w, const uint16_t* w_lo,
                         const uint16_t* w16, int ldw, const long long* targets, long long R, int C, int K, float grad_scale, float* row_loss, float* loss,
                         uint16_t* dlogits, int ld_dl, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!h || !w || !targets || !row_loss || !loss) return AMDNUWA_ERR_ARG;
    if (C % 64 || K % 32 || ldh % 8 || ldw % 8 || (dlogits && ld_dl % 8) || R > 0x7fffffffLL) return AMDNUWA_ERR_UNSUPPORTED;
    if (R <= 0) return AMDNUWA_OK;
    if (!workspace || workspace_bytes < amdnuwa_linear_ce_workspace_bytes(R, C)) return AMDNUWA_ERR_WORKSPACE;
    const bool x3 = h_lo != nullptr;
    const int nblk = C / 64;
    float* stats = (float*)workspace;
    float* lse = stats + (size_t)R * nblk * 2;
    float* tl = lse + R;
    hipError_t e = hipMemsetAsync(tl, 0xff, (size_t)R * sizeof(float), stream);          // NaN until a valid target column writes it
    if (e != hipSuccess) return (int)e;
    GemmArgs p{};
    p.A = (const bf16_t*)h; p.Alo = (const bf16_t*)h_lo; p.lda = ldh; p.B = (const bf16_t*)w; p.Blo = (const bf16_t*)w_lo; p.ldb = ldw;
    p.C = dlogits; p.ldc = ld_dl; p.alpha = 1.f;
    p.M = (int)R; p.N = C; p.K = K; p.shift_dim = K;
    p.tiles_m = (int)((R + 255) / 256); p.tiles_n = (C + 255) / 256;
    p.ce_stats = stats; p.ce_tl = tl; p.ce_lse = lse; p.ce_tgt = targets; p.ce_scale = grad_scale; p.ce_nblk = nblk;
    p.skew = nt_skew((long long)p.tiles_m * p.tiles_n);
    dim3 grid(p.tiles_m * p.tiles_n, 1), block(512);
    const size_t lds = x3 ? (size_t)2 * 4 * 256 * 32 * 2 : (size_t)4 * 2 * 256 * 32 * 2;
    if (x3) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_nt_256x3_kernel<2>), grid, block, lds, stream, p);
    } else {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel<false, 2, 4, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_nt_256_kernel<false, 2, 4, 4, 1>), grid, block, lds, stream, p);
    }
    LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_combine_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, stats, tl, nblk, R, lse, row_loss);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(1024), 0, stream, row_loss, R, loss);
    LAUNCH_CHECK();
    if (dlogits) {
        if (x3 && h16 && w16) {
            p.A = (const bf16_t*)h16; p.B = (const bf16_t*)w16; p.Alo = nullptr; p.Blo = nullptr;
            const size_t l16 = (size_t)4 * 2 * 256 * 32 * 2;
            (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel<false, 3, 4, 4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16);
            hipLaunchKernelGGL((gemm_nt_256_kernel<false, 3, 4, 4, 1, true>), grid, block, l16, stream, p);
        } else if (x3) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_nt_256x3_kernel<3>), grid, block, lds, stream, p);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel<false, 3, 4, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_nt_256_kernel<false, 3, 4, 4, 1>), grid, block, lds, stream, p);
        }
        LAUNCH_CHECK();
    }
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_linear_ce(const uint16_t* h, int ldh, const uint16_t* w, int ldw, const long long* targets, long long R, int C,
                                 int K, float grad_scale, float* row_loss, float* loss, uint16_t* dlogits, int ld_dl,
                                 void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return linear_ce_run(h, nullptr, nullptr, ldh, w, nullptr, nullptr, ldw, targets, R, C, K, grad_scale, row_loss, loss, dlogits, ld_dl,
                         workspace, workspace_bytes, stream);
}

extern "C" int amdnuwa_linear_ce_x3(const uint16_t* h_hi, const uint16_t* h_lo, const uint16_t* h_f16, int ldh, const uint16_t* w_hi,
                                    const uint16_t* w_lo, const uint16_t* w_f16, int ldw, const long long* targets, long long R, int C, int K,
                                    float grad_scale, float* row_loss, float* loss, uint16_t* dlogits, int ld_dl, void* workspace,
                                    size_t workspace_bytes, hipStream_t stream) {
    if (!h_lo || !w_lo || ((h_f16 != nullptr) != (w_f16 != nullptr))) return AMDNUWA_ERR_ARG;
    return linear_ce_run(h_hi, h_lo, h_f16, ldh, w_hi, w_lo, w_f16, ldw, targets, R, C, K, grad_scale, row_loss, loss, dlogits, ld_dl, workspace,
                         workspace_bytes, stream);
}

extern "C" int amdnuwa_gemm_nt_fused(const amdnuwa_gemm_desc* d) { return d && d->C2 && nt_geglu_fusable(d) ? 1 : 0; }

// fp16 operands (d->ab_f16): the 256x256 ring only -- the FeedForward GEMMs of the 'bf16x3-fwd' forward
extern "C" int amdnuwa_gemm_nt_f16ops_supported(const amdnuwa_gemm_desc* d) {
    if (!d || !d->A || !d->B || !d->C || d->Alo || d->Blo || d->shift_ntok > 0 || d->batch > 1) return 0;
    if (d->geglu_u && !(d->c_f16 && d->c_is_bf16 && d->C2 && !d->Clo && !d->C2lo && !d->geglu_u_lo && d->ld_u % 8 == 0)) return 0;   // GEGLU backward: fp16 du only
    if (d->c_f16 && (!d->c_is_bf16 || d->Clo || d->C2lo || (d->C2 && !d->geglu_u))) return 0;     // fp16 output: one copy, no forward gate
    if (d->Clo && (!d->c_is_bf16 || d->C2)) return 0;              // Clo = fp16 copy of a bf16 output (no gate at the same time)
    if (d->K % 32 || d->lda % 8 || d->ldb % 8 || d->M <= 4 * ROWS_MR) return 0;
    if (d->c_is_bf16 && (d->N % 16 || d->ldc % 8 || (d->C2 && d->ldc2 % 8))) return 0;
    if (!d->c_is_bf16 && (d->C2 || d->N % 4 || d->ldc % 4)) return 0;
    // (no tile-count threshold above M = 4 * ROWS_MR = 32 rows: the arithmetic of a FeedForward block must not depend on the batch size -- a
    //  one-sample parity check (M = n = 2560) has to run the same fp16 products as the training batch, so small M takes the 256x256 ring too.
    //  M <= 32 -- the few-row steps of generate() -- stays on the hi + lo row kernel: more exact, and documented in DESIGN.md section 3)
    const int v = g_amdnuwa_tuning[0];
    return (v == 0 || v == 7 || v == 6 || v == 10 || v == 11) ? 1 : 0;          // (6: the 256x128 two-workgroups-per-CU probe of the same ring; 10: the K-step 64 form)
}

// fp16 A x fp16 (hi + lo) B, two MFMAs per product (d->ab_f16 with Blo): the hi + lo ring's X2 form.  As for the one-MFMA fp16 products
// there is no tile-count threshold above M = 32 rows -- the arithmetic of a block must not depend on the batch size; the few-row decode
// steps (M <= 4 * ROWS_MR) run the more exact hi + lo row kernel instead.
extern "C" int amdnuwa_gemm_nt_f16x2_supported(const amdnuwa_gemm_desc* d) {
    if (!d || !d->A || !d->B || !d->Blo || !d->C || d->Alo || d->shift_ntok > 0 || d->batch > 1 || d->C2 || d->geglu_u) return 0;
    if (d->K % 32 || d->lda % 8 || d->ldb % 8 || d->M <= 4 * ROWS_MR) return 0;
    if (d->c_is_bf16 ? (d->N % 8 || d->ldc % 8) : d->Clo != nullptr) return 0;
    if (d->c_f16 && (!d->c_is_bf16 || d->Clo)) return 0;          // one fp16 output: a 16-bit C, no second copy
    const int v = g_amdnuwa_tuning[0];
    return (v == 0 || v == 7) ? 1 : 0;
}

// does this product run on the bf16x3 256x256 ring (the only kernel that writes the fp16 second copy)?
static bool nt_x3_ring(const amdnuwa_gemm_desc* d) {
    if (!d->Alo || !d->Blo || d->shift_ntok > 0 || d->K % 32 || g_amdnuwa_tuning[13] == 1) return false;
    const int v = g_amdnuwa_tuning[0];
    if (v == 7) return true;
    if (v != 0 || d->M <= 4 * ROWS_MR) return false;
    return (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * (d->batch > 0 ? d->batch : 1) >= 512;
}
extern "C" int amdnuwa_gemm_nt_f16_fused(const amdnuwa_gemm_desc* d) {
    return d && d->c_is_bf16 && d->Clo && !d->C2 && nt_x3_ring(d) && d->N % 8 == 0 && d->ldc % 8 == 0 ? 1 : 0;
}

extern "C" int amdnuwa_gemm_nt(const amdnuwa_gemm_desc* d, hipStream_t stream) {
    if (!d || !d->A || !d->B || !d->C) return AMDNUWA_ERR_ARG;
    if (d->M <= 0 || d->N <= 0) return AMDNUWA_OK;
    if (d->geglu_u && !d->C2) return AMDNUWA_ERR_ARG;
    if (d->C2 && !d->ab_f16 && !nt_geglu_fusable(d)) {             // plain product, then the stand-alone gate kernel on the same layout
        if (!d->c_is_bf16 || d->ldc != d->N || d->batch > 1) return AMDNUWA_ERR_ARG;
        amdnuwa_gemm_desc plain = *d;
        plain.C2 = nullptr; plain.C2lo = nullptr; plain.geglu_u = nullptr; plain.geglu_u_lo = nullptr;
        const int rc = amdnuwa_gemm_nt(&plain, stream);
        if (rc) return rc;
        if (d->geglu_u) {
            if (d->N % 8 || d->ld_u != 2 * d->N || d->ldc2 != 2 * d->N) return AMDNUWA_ERR_ARG;
            return amdnuwa_geglu_il_bwd(d->geglu_u, d->geglu_u_lo, (const uint16_t*)d->C, d->Clo, d->C2, d->C2lo, d->M, d->N, stream);
        }
        if (d->N % 16 || d->ldc2 != d->N / 2) return AMDNUWA_ERR_ARG;
        return amdnuwa_geglu_il_fwd((const uint16_t*)d->C, d->Clo, d->C2, d->C2lo, d->M, d->N / 2, stream);
    }
    if (d->ab_f16 && d->Blo) {                 // fp16 activation x fp16 (hi + lo) weight: two MFMAs per product on the hi + lo ring
        if (!amdnuwa_gemm_nt_f16x2_supported(d)) return AMDNUWA_ERR_UNSUPPORTED;
        GemmArgs q{};
        q.A = (const bf16_t*)d->A; q.lda = d->lda; q.B = (const bf16_t*)d->B; q.Blo = (const bf16_t*)d->Blo; q.ldb = d->ldb;
        q.C = d->C; q.Clo = (bf16_t*)d->Clo; q.ldc = d->ldc; q.bias = d->bias; q.alpha = d->alpha;
        q.M = d->M; q.N = d->N; q.K = d->K; q.shift_dim = d->K;
        q.tiles_m = (d->M + 255) / 256; q.tiles_n = (d->N + 255) / 256;
        q.dbg = g_amdnuwa_tuning[7];
        q.lo_f16 = d->Clo ? 1 : 0;             // the second copy of a bf16 output is its fp16 rendering (the next fp16 consumer's operand)
        q.c_f16 = d->c_f16 ? 1 : 0;            // ... or C itself is the fp16 rendering and there is no other copy (fp16-gradient backward of the block)
        q.skew = nt_skew((long long)q.tiles_m * q.tiles_n);
        dim3 g3(q.tiles_m * q.tiles_n, 1), b3(512);
        const size_t l3 = (size_t)2 * 3 * 256 * 32 * 2;
        if (d->c_is_bf16) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256x3_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
            hipLaunchKernelGGL((gemm_nt_256x3_kernel<1, true>), g3, b3, l3, stream, q);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256x3_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
            hipLaunchKernelGGL((gemm_nt_256x3_kernel<0, true>), g3, b3, l3, stream, q);
        }
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    if (d->ab_f16) {
        if (!amdnuwa_gemm_nt_f16ops_supported(d)) return AMDNUWA_ERR_UNSUPPORTED;
        GemmArgs q{};
        q.A = (const bf16_t*)d->A; q.lda = d->lda; q.B = (const bf16_t*)d->B; q.ldb = d->ldb;
        q.C = d->C; q.Clo = (bf16_t*)d->Clo; q.ldc = d->ldc; q.bias = d->bias; q.alpha = d->alpha;
        q.M = d->M; q.N = d->N; q.K = d->K; q.shift_dim = d->K;
        q.tiles_m = (d->M + 255) / 256; q.tiles_n = (d->N + 255) / 256;
        q.dbg = g_amdnuwa_tuning[7];
        if (d->C2) { q.C2 = (bf16_t*)d->C2; q.C2lo = (bf16_t*)d->C2lo; q.ldc2 = d->ldc2; }
        if (d->geglu_u) { q.Uin = (const bf16_t*)d->geglu_u; q.ldu = d->ld_u; }
        q.c_f16 = d->c_f16 ? 1 : 0;
        q.skew = nt_skew((long long)q.tiles_m * q.tiles_n);
        // epilogue: 0 = fp32, 1 = bf16 (+ fp16 copy / forward gate), 4 = ONE fp16 output or the fp16 GEGLU backward (fp16-gradient backward)
        const int epi = d->c_f16 ? (d->geglu_u ? 5 : 4) : (d->c_is_bf16 ? 1 : 0);
#define F16_LAUNCH(KERNEL, GRID, BLOCK, LDS, ...)                                                                         \
    do {                                                                                                                  \
        if (epi == 4) {                                                                                                   \
            (void)hipFuncSetAttribute((const void*)KERNEL<__VA_ARGS__ 4 F16_TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ 4 F16_TAIL>), GRID, BLOCK, LDS, stream, q);                            \
        } else if (epi == 5) {                                                                                            \
            (void)hipFuncSetAttribute((const void*)KERNEL<__VA_ARGS__ EPI5 F16_TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ EPI5 F16_TAIL>), GRID, BLOCK, LDS, stream, q);                         \
        } else if (epi == 1) {                                                                                            \
            (void)hipFuncSetAttribute((const void*)KERNEL<__VA_ARGS__ 1 F16_TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ 1 F16_TAIL>), GRID, BLOCK, LDS, stream, q);                            \
        } else {                                                                                                          \
            (void)hipFuncSetAttribute((const void*)KERNEL<__VA_ARGS__ 0 F16_TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__ 0 F16_TAIL>), GRID, BLOCK, LDS, stream, q);                            \
        }                                                                                                                 \
        LAUNCH_CHECK();                                                                                                   \
        return AMDNUWA_OK;                                                                                                \
    } while (0)
        // (EPI5: the GEGLU-backward epilogue exists on the persistent ring and the plain ring only; elsewhere the name stands for EPI 4 and is never reached)
#define EPI5 4
        const size_t l64 = (size_t)2 * 2 * 256 * 64 * 2;
        dim3 g2(q.tiles_m * q.tiles_n, 1), b2(512);
        if (epi != 5 && (g_amdnuwa_tuning[0] == 11 || (g_amdnuwa_tuning[0] == 0 && d->K >= 1024 && d->K < 2048))) {     // K-step 64 form with staggered wave rows (FF2: -9 %)
#define F16_TAIL , 1, 4, 4, true
            F16_LAUNCH(gemm_nt_256_kernel, g2, b2, l64, false,);
#undef F16_TAIL
        }
        const size_t l2 = (size_t)4 * 2 * 256 * 32 * 2;
        if (epi != 5 && nt_long_k(d->K, q.dbg) && (g_amdnuwa_tuning[0] == 0 || g_amdnuwa_tuning[0] == 7)) { // long K: four waves of 128x128, K-step 64 (gemm_nt_w4k_kernel)
            dim3 b4(256);
#define F16_TAIL , true
            F16_LAUNCH(gemm_nt_w4k_kernel, g2, b4, l64, );
#undef F16_TAIL
        }
#undef EPI5
#define EPI5 5
        if (nt_persistent(q.tiles_m * q.tiles_n, d->K, q.dbg)) {               // one workgroup per CU walks the tile list (gemm_nt_256p_kernel)
            dim3 gp(nt_persistent_grid(), 1);
            const size_t l2p = l2 + (epi == 5 ? (size_t)8 * 4096 : 0);             // EPI 5: the waves' u tiles behind the ring (160 KiB in all)
#define F16_TAIL , true
            F16_LAUNCH(gemm_nt_256p_kernel, gp, b2, l2p, );
#undef F16_TAIL
        }
#define F16_TAIL , 4, 4, 1, true
        F16_LAUNCH(gemm_nt_256_kernel, g2, b2, l2, false,);
#undef F16_TAIL
#undef EPI5
#undef F16_LAUNCH
    }
    if (d->K % 8 || d->lda % 8 || d->ldb % 8) return AMDNUWA_ERR_ARG;
    if ((d->Alo == nullptr) != (d->Blo == nullptr)) return AMDNUWA_ERR_ARG;
    if (d->shift_ntok > 0 && (d->shift_fmap <= 0 || d->K % 32)) return AMDNUWA_ERR_ARG;
    if (d->c_lo_f16 && !amdnuwa_gemm_nt_f16_fused(d)) return AMDNUWA_ERR_UNSUPPORTED;
    GemmArgs p;
    p.A = (const bf16_t*)d->A; p.Alo = (const bf16_t*)d->Alo; p.sA = d->strideA; p.lda = d->lda;
    p.B = (const bf16_t*)d->B; p.Blo = (const bf16_t*)d->Blo; p.sB = d->strideB; p.ldb = d->ldb;
    p.C = d->C; p.Clo = (bf16_t*)d->Clo; p.sC = d->strideC; p.ldc = d->ldc;
    p.bias = d->bias; p.alpha = d->alpha; p.beta = 0.f;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.shift_ntok = d->shift_ntok; p.shift_fmap = d->shift_fmap; p.shift_dim = d->K;
    p.tiles_m = (d->M + BM - 1) / BM; p.tiles_n = (d->N + BN - 1) / BN;
    p.ksplit_len = 0;
    p.dbg = g_amdnuwa_tuning[7];
    p.skew = 0;
    p.C2 = nullptr; p.C2lo = nullptr; p.ldc2 = 0; p.Uin = nullptr; p.ldu = 0; p.lo_f16 = 0; p.c_f16 = 0;
    p.batch_inner = d->batch_inner; p.sA_in = d->strideA_inner; p.sB_in = d->strideB_inner; p.sC_in = d->strideC_inner;
    const bool x3 = d->Alo != nullptr, sh = d->shift_ntok > 0, ob = d->c_is_bf16 != 0;
    // a handful of rows (the decode step of generate()): stream the weight instead of running MFMA tiles
    if (g_amdnuwa_tuning[0] == 0 && d->M <= 4 * ROWS_MR && d->batch <= 1 && !sh && (size_t)ROWS_MR * d->K * 4 <= 144 * 1024) {
        if (d->Clo && !ob) return AMDNUWA_ERR_ARG;
        p.beta = (ob && !d->Clo) ? 1.f : 0.f;                  // destination-type marker for the rows kernel (see its epilogue)
        const size_t lds = (size_t)ROWS_MR * d->K * sizeof(float);
        dim3 g((d->N + 4 * ROWS_CPW - 1) / (4 * ROWS_CPW), (d->M + ROWS_MR - 1) / ROWS_MR);
        if (x3) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_nt_rows_kernel<true>), g, dim3(256), lds, stream, p);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_nt_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_nt_rows_kernel<false>), g, dim3(256), lds, stream, p);
        }
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    dim3 grid(p.tiles_m * p.tiles_n, d->batch > 0 ? d->batch : 1), block(256);
    // direct-to-LDS variants (tuning key 0: 0 = register-staged, 1 = glds BK 64, 2 = glds BK 32)
    // tuning key 0: 0 = auto, 1 = direct-to-LDS BK 64, 2 = direct-to-LDS BK 32, 3 / 4 = 256x256 tile with a 4- / 3-stage
    // DMA ring, 5 = register-staged.  auto: 256x256 ring when it yields >= 2 full rounds of tiles on 256 CUs, else BK 32.
    int variant = g_amdnuwa_tuning[0];
    if (variant == 0) {
        const long long t256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * (d->batch > 0 ? d->batch : 1);
        variant = (d->K % 32 == 0) ? (t256 >= 512 ? 7 : 2) : 5;
    }
    // bf16x3 on the 256x256 tile (2-stage ring of hi + lo images); tuning key 13 = 1 keeps the first-generation 128x128 kernel
    if (x3 && !sh && variant == 7 && d->K % 32 == 0 && g_amdnuwa_tuning[13] != 1) {
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        dim3 g3(p.tiles_m * p.tiles_n, d->batch > 0 ? d->batch : 1), b3(512);
        const size_t l3 = (size_t)2 * 4 * 256 * 32 * 2;
        if (d->C2) { p.C2 = (bf16_t*)d->C2; p.C2lo = (bf16_t*)d->C2lo; p.ldc2 = d->ldc2; }
        p.lo_f16 = (ob && d->c_lo_f16 && d->Clo) ? 1 : 0;
        p.skew = nt_skew((long long)p.tiles_m * p.tiles_n);
        if (ob) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
            hipLaunchKernelGGL((gemm_nt_256x3_kernel<1>), g3, b3, l3, stream, p);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
            hipLaunchKernelGGL((gemm_nt_256x3_kernel<0>), g3, b3, l3, stream, p);
        }
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    if (!x3 && !sh && variant == 12 && d->K % 64 == 0 && !d->C2 && !d->Clo && !(ob && d->bias)) {   // 4 waves x 128x128, K-step 64, 1.5-iteration prefetch
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        dim3 g4(p.tiles_m * p.tiles_n, d->batch > 0 ? d->batch : 1), b4(256);
        const size_t l4 = (size_t)2 * 2 * 256 * 64 * 2;
        if (ob) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_w4k_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
            hipLaunchKernelGGL((gemm_nt_w4k_kernel<1, false>), g4, b4, l4, stream, p);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_nt_w4k_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
            hipLaunchKernelGGL((gemm_nt_w4k_kernel<0, false>), g4, b4, l4, stream, p);
        }
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    if (variant == 12 || variant == 9 || variant == 8 || variant == 6 || variant == 3 || variant == 4) variant = 7;   // (3 / 4 / 6 / 8 / 9: probe variants of rounds 2-4, removed in round 6)
    if (variant == 1) variant = 2;
    // K-step 64 form with staggered wave rows (two 64 KiB stages of full 128-byte row lines): measured ahead of the K-step 32 ring on the
    // K = 1376 / 1536 shapes only (FF2 -10 %, dgrad qkv -1.5 %; K = 512 shapes and K >= 2752 equal or slower: profiles/r04g_gemm_k64.txt), so `auto`
    // takes it for 1024 <= K < 2048; tuning key 0 = 11 forces it, 7 keeps the ring
    if (!x3 && !sh && (variant == 11 || (g_amdnuwa_tuning[0] == 0 && variant == 7 && d->K >= 1024 && d->K < 2048 && !nt_long_k(d->K, p.dbg))) && d->K % 32 == 0 && !d->C2) {
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        dim3 g2(p.tiles_m * p.tiles_n, d->batch > 0 ? d->batch : 1), b2(512);
        const size_t l6 = (size_t)2 * 2 * 256 * 64 * 2;
        if (ob) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel<false, 1, 1, 4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l6);
            hipLaunchKernelGGL((gemm_nt_256_kernel<false, 1, 1, 4, 4, false>), g2, b2, l6, stream, p);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel<false, 0, 1, 4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l6);
            hipLaunchKernelGGL((gemm_nt_256_kernel<false, 0, 1, 4, 4, false>), g2, b2, l6, stream, p);
        }
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    if (variant == 11) variant = 7;
    if (!x3 && variant == 7 && d->K % 32 == 0) {                           // 256x256 tile, 4-stage ring, staggered wave rows
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        if (d->C2) { p.C2 = (bf16_t*)d->C2; p.ldc2 = d->ldc2; p.Uin = (const bf16_t*)d->geglu_u; p.ldu = d->ld_u; }
        p.skew = nt_skew((long long)p.tiles_m * p.tiles_n);
        dim3 g2(p.tiles_m * p.tiles_n, d->batch > 0 ? d->batch : 1), b2(512);
        if (!sh && nt_long_k(d->K, p.dbg)) {                                   // long K: four waves of 128x128, K-step 64 (gemm_nt_w4k_kernel)
            const size_t l4 = (size_t)2 * 2 * 256 * 64 * 2;
            dim3 b4(256);
            if (ob) {
                (void)hipFuncSetAttribute((const void*)gemm_nt_w4k_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
                hipLaunchKernelGGL((gemm_nt_w4k_kernel<1, false>), g2, b4, l4, stream, p);
            } else {
                (void)hipFuncSetAttribute((const void*)gemm_nt_w4k_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
                hipLaunchKernelGGL((gemm_nt_w4k_kernel<0, false>), g2, b4, l4, stream, p);
            }
            LAUNCH_CHECK();
            return AMDNUWA_OK;
        }
        if (!sh && nt_persistent(p.tiles_m * p.tiles_n, d->K, p.dbg)) {        // one workgroup per CU walks the tile list (gemm_nt_256p_kernel)
            const size_t l2 = (size_t)4 * 2 * 256 * 32 * 2;
            dim3 gp(nt_persistent_grid(), d->batch > 0 ? d->batch : 1);
            if (ob) {
                (void)hipFuncSetAttribute((const void*)gemm_nt_256p_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
                hipLaunchKernelGGL((gemm_nt_256p_kernel<1, false>), gp, b2, l2, stream, p);
            } else {
                (void)hipFuncSetAttribute((const void*)gemm_nt_256p_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
                hipLaunchKernelGGL((gemm_nt_256p_kernel<0, false>), gp, b2, l2, stream, p);
            }
            LAUNCH_CHECK();
            return AMDNUWA_OK;
        }
#define GS(SH, EP)                                                                                                    \
    do {                                                                                                              \
        const size_t l2 = (size_t)4 * 2 * 256 * 32 * 2;                                                               \
        (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel<SH, EP, 4, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2); \
        hipLaunchKernelGGL((gemm_nt_256_kernel<SH, EP, 4, 4, 1>), g2, b2, l2, stream, p);                           \
    } while (0)
        if (sh) { if (ob) GS(true, 1); else GS(true, 0); } else { if (ob) GS(false, 1); else GS(false, 0); }
#undef GS
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    if (!x3 && variant == 2 && d->K % 32 == 0) {
        const size_t gl = (size_t)2 * 2 * 128 * 32 * 2;
#define GL_LAUNCH(BK__, SH, EP) hipLaunchKernelGGL((gemm_nt_glds_kernel<BK__, SH, EP>), grid, block, gl, stream, p)
        if (sh) { if (ob) GL_LAUNCH(32, true, 1); else GL_LAUNCH(32, true, 0); } else { if (ob) GL_LAUNCH(32, false, 1); else GL_LAUNCH(32, false, 0); }
#undef GL_LAUNCH
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    const size_t lds = (size_t)2 * (x3 ? 4 : 2) * TILE_BYTES;
#define NT_LAUNCH(X3, SH, EP)                                                                                      \
    do {                                                                                                             \
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<X3, SH, EP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_nt_kernel<X3, SH, EP>), grid, block, lds, stream, p);                               \
    } while (0)
    if (x3) { if (sh) { if (ob) NT_LAUNCH(true, true, 1); else NT_LAUNCH(true, true, 0); } else { if (ob) NT_LAUNCH(true, false, 1); else NT_LAUNCH(true, false, 0); } }
    else    { if (sh) { if (ob) NT_LAUNCH(false, true, 1); else NT_LAUNCH(false, true, 0); } else { if (ob) NT_LAUNCH(false, false, 1); else NT_LAUNCH(false, false, 0); } }
#undef NT_LAUNCH
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

// tuning key 6: 0 = auto (256x256 4-stage ring when both output dims >= 256, else direct-to-LDS 128x128),
//               1 = register-staged 128x128, 2 = direct-to-LDS 128x128, 3 = 256x256 ring
static int tn_variant(const amdnuwa_gemm_desc* d) {
    const bool x3 = d->Alo != nullptr, big = (long long)d->K >= (1LL << 31);
    if (x3 || big) return 1;
    int v = g_amdnuwa_tuning[6];
    if (v == 0) v = (d->M >= 256 && d->N >= 256) ? 3 : 2;
    if (v == 3 && !(d->M >= 256 && d->N >= 256)) v = 2;
    return v;
}
// the four-wave weight-gradient kernel (gemm_tn_w4k_kernel): plain bf16 operands, no token shift, token rows and column counts it can cover
// without edge handling (tuning key 23 = 1 keeps the 8-wave ring)
static bool tn_w4k_ok(const amdnuwa_gemm_desc* d) {
    if (g_amdnuwa_tuning[23] == 1 || d->Alo || d->shift_ntok > 0) return false;
    return d->K % 64 == 0 && d->K >= 128 && d->M >= 256 && d->N >= 256;
}
// the whole-M narrow kernel (gemm_tn_wm_kernel): batched, N <= 64, 128 < M <= 384 (tuning key 25 = 1 keeps the 128-row tiles).  Returns MT or 0.
static int tn_whole_m(const amdnuwa_gemm_desc* d) {
    if (g_amdnuwa_tuning[25] == 1 || d->Alo || d->shift_ntok > 0 || (long long)d->K >= (1LL << 31)) return 0;
    const int v = g_amdnuwa_tuning[6];
    if ((v != 0 && v != 2) || d->N > 64 || d->N < 1 || d->M <= 128 || d->M > 384 || d->batch < 2) return 0;
    return (d->M + 127) / 128;
}
// split-K policy: fill the workgroup SLOTS of the chip exactly once (256 CUs x resident workgroups per CU);
// never exceed them (a 257th workgroup would cost a whole extra round), keep >= minrows token rows per split.
static int tn_splits(const amdnuwa_gemm_desc* d) {
    const int v = tn_variant(d);
    const int tl = v == 3 ? 256 : 128;
    const bool wm = tn_whole_m(d) != 0;                      // one workgroup per batch element, one workgroup per CU
    const int tiles = wm ? d->batch : ((d->M + tl - 1) / tl) * ((d->N + tl - 1) / tl) * (d->batch > 0 ? d->batch : 1);
    const int slots = g_amdnuwa_tuning[1] > 0 ? g_amdnuwa_tuning[1] : (v == 3 || wm ? 256 : 1024);
    const int minrows = g_amdnuwa_tuning[2] > 0 ? g_amdnuwa_tuning[2] : 256;
    int splits = slots / tiles;                              // floor: stay within one round
    const int maxs = (d->K + minrows - 1) / minrows;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    return splits;
}

// A in planes of 32 columns (d->a_chunk32): the whole-M narrow kernel only
extern "C" int amdnuwa_gemm_tn_chunked_a_supported(const amdnuwa_gemm_desc* d) {
    if (!d || d->lda % 8 || d->ldb % 8) return 0;
    return tn_variant(d) == 2 && tn_whole_m(d) != 0 ? 1 : 0;
}

// fp16 operands (d->ab_f16): the four-wave kernel and the whole-M narrow kernel only
extern "C" int amdnuwa_gemm_tn_f16_supported(const amdnuwa_gemm_desc* d) {
    if (!d || d->Alo || d->Blo || d->shift_ntok > 0 || d->lda % 8 || d->ldb % 8) return 0;
    if (tn_variant(d) == 2) return tn_whole_m(d) != 0 ? 1 : 0;
    return tn_variant(d) == 3 && tn_w4k_ok(d) ? 1 : 0;
}

extern "C" size_t amdnuwa_gemm_tn_workspace_bytes(const amdnuwa_gemm_desc* d) {
    if (!d) return 0;
    return (size_t)tn_splits(d) * (d->batch > 0 ? d->batch : 1) * (size_t)d->M * d->N * sizeof(float);
}

// C[N1=M, N2=N] (fp32) = beta*C + alpha * A[K rows, M]^T . B[K rows, N]; shift (if any) applies to B.
extern "C" int amdnuwa_gemm_tn(const amdnuwa_gemm_desc* d, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!d || !d->A || !d->B || !d->C || d->c_is_bf16) return AMDNUWA_ERR_ARG;
    if (d->M <= 0 || d->N <= 0) return AMDNUWA_OK;
    if (d->lda % 8 || d->ldb % 8) return AMDNUWA_ERR_ARG;   // operand rows must be readable up to the next multiple of 8 columns
    if ((d->Alo == nullptr) != (d->Blo == nullptr)) return AMDNUWA_ERR_ARG;
    if (d->shift_ntok > 0 && (d->shift_fmap <= 0 || d->N % 32)) return AMDNUWA_ERR_ARG;
    if (workspace_bytes < amdnuwa_gemm_tn_workspace_bytes(d) || !workspace) return AMDNUWA_ERR_WORKSPACE;
    if (d->ab_f16 && !amdnuwa_gemm_tn_f16_supported(d)) return AMDNUWA_ERR_UNSUPPORTED;
    if (d->a_chunk32 && !amdnuwa_gemm_tn_chunked_a_supported(d)) return AMDNUWA_ERR_UNSUPPORTED;
    GemmArgs p;
    p.c_f16 = 0; p.a_chunk = d->a_chunk32 ? 1 : 0; p.alpha_dev = d->alpha_dev;
    p.A = (const bf16_t*)d->A; p.Alo = (const bf16_t*)d->Alo; p.sA = d->strideA; p.lda = d->lda;
    p.B = (const bf16_t*)d->B; p.Blo = (const bf16_t*)d->Blo; p.sB = d->strideB; p.ldb = d->ldb;
    p.C = d->C; p.Clo = nullptr; p.sC = d->strideC; p.ldc = d->ldc;
    p.bias = nullptr; p.alpha = d->alpha; p.beta = d->beta;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.shift_ntok = d->shift_ntok; p.shift_fmap = d->shift_fmap; p.shift_dim = d->N;
    p.tiles_m = (d->M + 127) / 128; p.tiles_n = (d->N + 127) / 128;
    const int splits = tn_splits(d);
    int len = (d->K + splits - 1) / splits;
    len = (len + TK - 1) / TK * TK;
    if (tn_variant(d) == 3 && d->shift_ntok <= 0 && tn_w4k_ok(d)) len = (len + 63) / 64 * 64;      // whole 64-row iterations per split
    p.ksplit_len = len;
    p.batch_inner = d->batch_inner; p.sA_in = d->strideA_inner; p.sB_in = d->strideB_inner; p.sC_in = d->strideC_inner;
    const int batch = d->batch > 0 ? d->batch : 1;
    const bool x3 = d->Alo != nullptr, sh = d->shift_ntok > 0;
    dim3 grid(p.tiles_m * p.tiles_n, batch, splits), block(256);
    const size_t lds = (size_t)2 * (x3 ? 4 : 2) * TN_TILE_BYTES;
    float* part = (float*)workspace;
    const int tnv = tn_variant(d);
    if (tnv == 3) {
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        p.nsplit = splits;
        dim3 g256(p.tiles_m * p.tiles_n * splits, batch, 1), b256(512);
        const size_t l256 = (size_t)4 * 2 * 32 * 256 * 2;
#define TN256(SH, ST)                                                                                                  \
    do {                                                                                                              \
        (void)hipFuncSetAttribute((const void*)gemm_tn_256_kernel<SH, 4, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l256); \
        hipLaunchKernelGGL((gemm_tn_256_kernel<SH, 4, ST>), g256, b256, l256, stream, p, part);                        \
    } while (0)
        // tuning key 8: 0 = auto = lock-step, 1 = lock-step, 2 = staggered wave rows.  With the DMA pieces issued through asm (common.h:
        // the compiler no longer drains the ring before the fragment reads) the lock-step loop gained 15-27 % and passed the staggered
        // one on every weight-gradient shape of the step (r02: dW qkv 466 -> 340 us vs 424 staggered; dW ff1 787 -> 608 vs 773)
        const bool stag = g_amdnuwa_tuning[8] == 2;
        if (!sh && tn_w4k_ok(d)) {                                               // four waves, 64 token rows per iteration (gemm_tn_w4k_kernel)
            dim3 b4(256);
            const size_t l4 = (size_t)2 * 2 * 64 * 256 * 2;
            if (d->ab_f16) {
                (void)hipFuncSetAttribute((const void*)gemm_tn_w4k_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
                hipLaunchKernelGGL(gemm_tn_w4k_kernel<true>, g256, b4, l4, stream, p, part);
            } else {
                (void)hipFuncSetAttribute((const void*)gemm_tn_w4k_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
                hipLaunchKernelGGL(gemm_tn_w4k_kernel<false>, g256, b4, l4, stream, p, part);
            }
        } else
        if (sh) { if (stag) TN256(true, true); else TN256(true, false); }
        else    { if (stag) TN256(false, true); else TN256(false, false); }
#undef TN256
    } else
    if (tnv == 2 && tn_whole_m(d) != 0) {
        const int mt = tn_whole_m(d);
        const dim3 gw(batch, splits);
        const bool direct = splits == 1 && d->beta == 0.f && g_amdnuwa_tuning[25] != 2;      // (a device factor rides in the direct epilogue too)
        p.nsplit = direct ? 0 : splits;
#define TNWM(MT_, F_)                                                                                                    \
    do {                                                                                                                 \
        const size_t l = (size_t)4 * (MT_ + 1) * TN_TILE_BYTES;                                                          \
        (void)hipFuncSetAttribute((const void*)gemm_tn_wm_kernel<MT_, 4, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
        hipLaunchKernelGGL((gemm_tn_wm_kernel<MT_, 4, F_>), gw, block, l, stream, p, part);                               \
    } while (0)
#define TNWMF(MT_, MI_, F_)                                                                                              \
    do {                                                                                                                 \
        const size_t l = (size_t)4 * (MT_ + 1) * TN_TILE_BYTES;                                                          \
        (void)hipFuncSetAttribute((const void*)gemm_tn_wmf_kernel<MT_, MI_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
        hipLaunchKernelGGL((gemm_tn_wmf_kernel<MT_, MI_, F_>), gw, block, l, stream, p, part);                            \
    } while (0)
        // the lean form: whole K-steps in every split, 32-bit piece offsets (tuning key 25 = 3 keeps the general kernel)
        const bool lean = g_amdnuwa_tuning[25] != 3 && d->K % TK == 0 && (long long)((d->M + 31) / 32) * d->K * 64 < (1LL << 32) &&
                          32LL * d->lda * 2 < (1LL << 31) && 32LL * d->ldb * 2 < (1LL << 31);
        const int mi = ((d->M + 15) / 16 + 3) / 4;                                // row fragments per wave
        if (lean) {
            if (mt == 3) { if (mi == 5) { if (d->ab_f16) TNWMF(3, 5, true); else TNWMF(3, 5, false); } else { if (d->ab_f16) TNWMF(3, 6, true); else TNWMF(3, 6, false); } }
            else         { if (mi == 3) { if (d->ab_f16) TNWMF(2, 3, true); else TNWMF(2, 3, false); } else { if (d->ab_f16) TNWMF(2, 4, true); else TNWMF(2, 4, false); } }
        } else
        if (mt == 3) { if (d->ab_f16) TNWM(3, true); else TNWM(3, false); }
        else         { if (d->ab_f16) TNWM(2, true); else TNWM(2, false); }
#undef TNWMF
#undef TNWM
    } else
    if (tnv == 2) {
        const size_t gl = (size_t)2 * 2 * TN_TILE_BYTES;
        if (sh) hipLaunchKernelGGL((gemm_tn_glds_kernel<true>), grid, block, gl, stream, p, part);
        else if (d->N <= 64) {
            const size_t gl4 = (size_t)4 * 2 * TN_TILE_BYTES;
            (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<false, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gl4);
            hipLaunchKernelGGL((gemm_tn_glds_kernel<false, true, 4>), dim3(grid.x * grid.y, 1, grid.z), block, gl4, stream, p, part);
        }
        else    hipLaunchKernelGGL((gemm_tn_glds_kernel<false>), grid, block, gl, stream, p, part);
    } else
    if (x3) { if (sh) hipLaunchKernelGGL((gemm_tn_kernel<true, true>), grid, block, lds, stream, p, part);
              else    hipLaunchKernelGGL((gemm_tn_kernel<true, false>), grid, block, lds, stream, p, part); }
    else    { if (sh) hipLaunchKernelGGL((gemm_tn_kernel<false, true>), grid, block, lds, stream, p, part);
              else    hipLaunchKernelGGL((gemm_tn_kernel<false, false>), grid, block, lds, stream, p, part); }
    LAUNCH_CHECK();
    if (tnv == 2 && tn_whole_m(d) != 0 && p.nsplit == 0) return AMDNUWA_OK;      // (the whole-M kernel wrote C itself)
    const size_t per = (size_t)d->M * d->N;
    int rb = (int)((per + 255) / 256); if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb, batch), dim3(256), 0, stream, part, (float*)d->C, (long long)d->strideC,
                       (long long)d->strideC_inner, d->batch_inner, d->ldc, d->M, d->N, splits, d->alpha, d->beta, d->alpha_dev);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

AMDNUWA_SAT_ACCESSOR(gemm)
