// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of libamdnuwa.
// Wave = 64 lanes everywhere.  bf16 values travel as raw uint16_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define AMDNUWA_OK 0
#define AMDNUWA_ERR_ARG -1
#define AMDNUWA_ERR_UNSUPPORTED -2
#define AMDNUWA_ERR_WORKSPACE -3
#define AMDNUWA_ERR_COMM -4

// runtime tuning knobs (amdnuwa_set_tuning): [0] NT GEMM variant, [1] TN target workgroups, [2] TN min rows per split
extern int g_amdnuwa_tuning[32];

#define LAUNCH_CHECK()                                \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (inputs are finite on this path)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// split fp32 into a bf16 hi part and a bf16 residual (hi + lo carries ~16 mantissa bits)
__device__ __forceinline__ void f2bf_hilo(float f, bf16_t& hi, bf16_t& lo) {
    hi = f2bf(f);
    lo = f2bf(f - bf2f(hi));
}

__device__ __forceinline__ uint32_t pack2(bf16_t a, bf16_t b) { return (uint32_t)a | ((uint32_t)b << 16); }
// two fp32 -> packed bf16 pair (a in the low half) with the gfx950 converter (v_cvt_pk_bf16_f32, round-to-nearest-even:
// bit-identical to f2bf on finite values, one instruction instead of eight)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2_rne(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ---- fp16 operand form of the attention cores in the 'bf16x3-fwd' mode -------------------------------------------------------
// The forward attention cores of that mode run SINGLE fp16 MFMAs (v_mfma_f32_16x16x32_f16: the bf16 rate, 11 significand bits
// instead of 8) on q / k / v stored as fp16 next to their bf16 copies; everything else about the kernels is the bf16 fast path.
// 16-bit payloads travel in the same bf16x8 / uint32 registers whatever their type; the helpers below pick the type by a
// template flag.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack2_f16(float a, float b) {          // v_cvt_pk_f16_f32, round to nearest even
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float f16lo_f(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float f16hi_f(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
// SATURATING forms for activation / weight / gradient stores (LayerNorm copies, GEMM epilogues, K / V images): fp16 ends at 65504 and
// the plain converter returns inf beyond it, which an MFMA turns into NaN rows.  The clamp is v_minimum3_f32 + v_maximum3_f32 (gfx950:
// the IEEE-754-2019 forms, which PROPAGATE NaN -- v_med3_f32 / v_min / v_max return the non-NaN operand, so a NaN activation used to become
// -65504 in the fp16 copy while its bf16 copy held NaN); bit-identical for every in-range value.  Probabilities (<= 1) keep the plain converter.
__device__ __forceinline__ float f16_clamp(float f) { return __builtin_elementwise_maximum(__builtin_elementwise_minimum(f, 65504.f), -65504.f); }
// (round 6: the PAIR is clamped after the conversion, on the packed halves -- v_pk_minimum3_f16 + v_pk_maximum3_f16, two instructions per pair
//  instead of four; the converter returns +-inf beyond the range and keeps NaN, the IEEE-754-2019 minimum / maximum turn inf into +-65504 and
//  propagate NaN: the same bits as clamping the fp32 values first, for every input)
__device__ __forceinline__ uint32_t pack2_f16_sat(float a, float b) {
    const f32x2_t v = {a, b};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f16x2_t top = {(_Float16)65504.f, (_Float16)65504.f}, bot = {(_Float16)-65504.f, (_Float16)-65504.f};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_minimum(h, top), bot));
}
__device__ __forceinline__ uint16_t f2h_sat(float f) { return f2h(f16_clamp(f)); }
// Saturation monitor: a store site keeps the running max |value| it handed to a saturating store (pack2_f16_sat_n: one v_max3_f32 per
// pair) and reports ONCE per thread at the end (f16_sat_commit: an atomic only when something actually saturated).  The counter is one
// word per translation unit (internal linkage); amdnuwa_f16_sat_count() sums the units (api.hip, AMDNUWA_SAT_ACCESSOR below).
namespace { __device__ unsigned int g_f16_sat_events = 0; }
__device__ __forceinline__ uint32_t pack2_f16_sat_n(float a, float b, float& amax) {
    amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
    return pack2_f16_sat(a, b);
}
__device__ __forceinline__ void f16_sat_commit(float amax) {
    if (amax > 65504.f) atomicAdd(&g_f16_sat_events, 1u);
}
#define AMDNUWA_SAT_ACCESSOR(NAME)                                                                                  \
    extern "C" __attribute__((visibility("hidden"))) unsigned amdnuwa_sat_##NAME(int reset) {                       \
        unsigned v = 0;                                                                                             \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_sat_events), sizeof(v)) != hipSuccess) return 0;               \
        if (reset && v) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f16_sat_events), &z, sizeof(z)); } \
        return v;                                                                                                   \
    }
// fp16 GRADIENTS of the 'bf16x3-fwd' backward (round 5): a gradient tensor travels as fp16(S * value) with S a power of two chosen once per
// backward pass from the largest magnitude of the residual-stream gradient entering it (S * amax in [8, 16): the headroom to 65504 covers the
// growth through a LayerNorm backward, the 11-bit significand keeps everything down to 2^-17 of that maximum normal).  Kernels receive a
// device pointer to {S, 1 / S} (NULL = no scaling) so that no host synchronisation is needed.
__device__ __forceinline__ float f16_gs(const float* scale2) { return scale2 ? scale2[0] : 1.f; }
__device__ __forceinline__ float f16_gs_inv(const float* scale2) { return scale2 ? scale2[1] : 1.f; }
template <bool F16> __device__ __forceinline__ uint32_t pack2_t(float a, float b) { return F16 ? pack2_f16(a, b) : pack2_rne(a, b); }
template <bool F16> __device__ __forceinline__ float lo_t(uint32_t u) { return F16 ? f16lo_f(u) : lo_f(u); }
template <bool F16> __device__ __forceinline__ float hi_t(uint32_t u) { return F16 ? f16hi_f(u) : hi_f(u); }
template <bool F16> __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One 1-KiB LDS-DMA piece (64 lanes x 16 bytes, lane l lands at lds + 16 l) issued through INLINE ASM.
// Why not __builtin_amdgcn_global_load_lds: the compiler tracks the builtin as a write to LDS and, unable to see the counted
// s_waitcnt vmcnt(N) + s_barrier that order a DMA ring, puts an `s_waitcnt vmcnt(0)` in front of every later LDS read it considers
// aliasing (every __builtin_amdgcn_ds_read_tr16_b64, every read through an address-space-3 pointer) -- found in the ISA of all TN
// GEMM and cross-attention loops: the ring was drained at every K-step, pieces issued a moment earlier included, so a "4-stage"
// ring prefetched nothing.  Use ONLY where the code orders the ring explicitly (counted vmcnt + barrier); a __syncthreads() does
// not wait for these.  `lds` must be wave-uniform.
typedef __attribute__((address_space(3))) void* lds_vptr_t;
// FENCE = false drops the "memory" clobber: the compiler may then move ordinary loads / stores across the piece (the counted-vmcnt
// asm statements keep theirs, and asm volatile statements and barriers keep their order among themselves).
template <bool FENCE = true>
__device__ __forceinline__ void dma16_asm(const void* gsrc, void* lds) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr_t)lds);
    if (FENCE) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "m0");
}

// 64 per-lane values v[0..63] -> lane l receives sum over all 64 lanes of v[l].  Halving butterfly: at distance `off` a lane keeps the
// half of its values whose index has that bit equal to its own lane bit and sends the other half: 32 + 16 + ... + 1 = 63 exchanges,
// all independent within a step, instead of 64 x 6 dependent ones for 64 separate wave_sum() calls (measured in the cross-attention
// backward: 384 serialized ds_bpermute round trips per workgroup).  Fixed order: deterministic.
__device__ __forceinline__ float wave_sum64_transposed(float (&v)[64], int lane) {
#pragma unroll
    for (int off = 32, cnt = 32; off >= 1; off >>= 1, cnt >>= 1) {
        const bool upper = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
            const float send = upper ? v[i] : v[i + cnt];
            const float keep = upper ? v[i + cnt] : v[i];
            v[i] = keep + __shfl_xor(send, off, 64);
        }
    }
    return v[0];
}

// Wave reductions: the four steps inside a row of 16 lanes are DPP operands of the add itself (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror: no LDS crossbar round trip), the two steps across rows are ds_bpermute.  Every lane ends with the result.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);           // row_half_mirror
    v += dpp_f<0x140>(v);           // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// erf-GELU (F.gelu default, np.py:255-258) and its derivative.  Phi(x) = 0.5 * erfc(-x / sqrt 2) through the Abramowitz-Stegun
// 7.1.26 rational form (|abs error| <= 1.5e-7, no cancellation on the negative side): 1 rcp + 1 exp + 5 fma instead of the
// ~60-instruction libm erff, which made the GEGLU kernels VALU-bound.  Returns Phi(x); e = exp(-x*x/2).
__device__ __forceinline__ float norm_cdf_f(float x, float& e) {
    // (round 6, 16 -> 13 instructions: w = |x| sqrt(log2(e) / 2) serves the exponent directly (e = exp2(-w w): no separate log2(e) multiply),
    //  the factor 1/2 sits in the polynomial's coefficients, and the two sides of x share q = Phi(|x|) - 1/2 = 1/2 - (1/2) erfc(|x| / sqrt 2):
    //  Phi(x) = 1/2 + copysign(q, x) -- one fma, one bit-field insert and one add instead of two multiplies, a compare, a subtract and a select.
    //  Same |abs error| (2.9e-7 against 3.0e-7 over [-9, 9]); the far negative tail is absolute, not relative, as the rational form's error is.)
    const float w = fabsf(x) * 0.8493218002880191f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.2727374808792225f, w, 1.f));   // 0.3275911 * (1 / sqrt 2) / 0.84932180
    e = __builtin_amdgcn_exp2f(-w * w);
    const float ph = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5307027145f, -0.7265760135f), 0.7107068705f), -0.142248368f), 0.127414796f);
    const float q = fmaf(-ph, e, 0.5f);
    return 0.5f + copysignf(q, x);
}
__device__ __forceinline__ float gelu_f(float x) { float e; return x * norm_cdf_f(x, e); }
__device__ __forceinline__ float gelu_grad_f(float x) { float e; const float c = norm_cdf_f(x, e); return fmaf(x * 0.3989422804014327f, e, c); }
// both at once: gelu(x) and gelu'(x)
__device__ __forceinline__ void gelu_both_f(float x, float& y, float& dy) {
    float e; const float c = norm_cdf_f(x, e);
    y = x * c; dy = fmaf(x * 0.3989422804014327f, e, c);
}

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b % 8): gives each XCD a
// contiguous slab of logical tile ids so neighbouring tiles share operand panels in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}
