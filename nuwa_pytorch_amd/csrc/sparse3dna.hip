// Fused causal 3-D nearby attention core (Sparse3DNA.forward np.py:488-608) for gfx950.
//
// The reference materialises unfolded K and V of shape (b*h, n, K+1, d) with unfoldNd
// (np.py:526-534); here nothing of size n*J*d exists.  One workgroup owns one query ROW of the
// token grid (all W queries (f, y, 0..W-1), all heads): for every causal tap plane (a, b) it
// stages the key row (f - (kf-1-a)df, y - (kh-1-b)dh, 0..W-1) of all heads in LDS once, and the
// kw taps along w of all W queries read it from there.  Scores for all (query, tap, head) of the
// row live in LDS (W*J*h fp32 <= 23.5 KiB), so the fp32 softmax and the talking-heads 8x8 mix
// (np.py:554-558) happen in-workgroup before the P.V pass.
//
// Thread map: t = ((w*NH + h)*4 + c): 4 lanes share one (query, head) and split the head dim in
// chunks of DH/4 (a key/value row of one query position is read as one contiguous 2*NH*DH bytes).
//
// Backward is split so that it needs no atomics and is bit-reproducible:
//   bwd_q  (query-centric, per query row): recompute P, dP' = dO.V, dP = W^T dP', ds, dq; writes
//          ds and P' (fp32, [B][nq][J][NH]) + per-workgroup partials for dW_th and the <bos> k/v.
//   bwd_kv (key-centric,  per key row): each key position PULLS from the <= K future queries that
//          attend to it (the transposed gather):  dk = scale * sum ds*q,  dv = sum P'*dO.
//   bwd_fin: fixed-order reduction of the partials (dW_th, dk[bos], dv[bos] + dO[bos]).
//
// hi/lo: every bf16 input may come with a bf16 residual (value = hi + lo); arithmetic is fp32 FMA.
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

struct S3Args {
    const bf16_t *q, *k, *v, *ql, *kl, *vl; int ld;       // q/k/v rows: [B*ntok, ld]
    bf16_t *o, *ol; int ldo;                              // fwd out
    int ol_f16;                                           // fp16 forward: ol receives the FP16 rendering of the output (the to_out GEMM's fp16 operand), not the bf16 residual
    const bf16_t *dO, *dOl; int lddo;                     // bwd in
    bf16_t *dq, *dk, *dv, *dql, *dkl, *dvl; int ldd;      // bwd out
    const float* wth;                                     // [NH][NH] talking heads (g, h)
    const float* bias;                                    // [J][NH] relative-position bias per key slot (or NULL)
    float *ds, *pm;                                       // [B][nq][J][NH]
    float* stats;                                         // recomputing key side (MFMA path): [B][nq][NH][4] = (row max, 1 / row sum, delta = sum_j P dP, -);
                                                          // non-NULL = the query side writes these INSTEAD of the ds / pm workspace
    float *part_th, *part_k0, *part_v0;                   // [B*F*H][NH*NH], [B*F*H][NH*DH] x2
    float* dwth;                                          // [NH*NH] accumulated
    int B, ntok, F, H, W, kf, kh, kw, df, dh, dw, NH;
    int of, oh, ow;                                       // tap index of the query's own position per axis: k - 1 (causal) / (k - 1) / 2 (symmetric)
    // key / value side.  Self-attention (Sparse3DNA): the query sequence itself, row 0 = <bos> = key slot 0.  xmode = 1 (SparseCross2DNA,
    // np.py:761-901): keys / values are a context grid of FK = kf frames, tap a of the frame axis IS context frame a (absolute), slot 0
    // is a learned null key / value, keys can be masked, and query row 0 (<bos>) is left to the host (it attends to everything).
    int xmode, FK, kvrows, kvoff, ldk, lddk;              // rows per sample / first grid row / row strides of the k, v (dk, dv) tensors
    const bf16_t *k0, *k0l, *v0, *v0l; long long k0_bs;   // slot-0 key / value rows [NH*DH] and their per-sample stride (elements)
    const uint8_t* kmask;                                 // [B][kvrows] (1 = visible) or NULL
    float *dnull_k, *dnull_v;                             // xmode: gradients of the null key / value [NH*DH] (fp32)
    float scale;
    int accumulate;
    int dbg;                                              // probe only (tuning key 9): bit 0 / 1 / 2 skip phase 1 / 2 / 3 of the MFMA forward
    int ymajor;                                           // MFMA kernels: workgroup order inside a sample is (y, f) instead of (f, y) (tuning key 3 bit 1 = old order)
    int sep_passes;                                       // MFMA query-side backward: the three separate item passes instead of the fused one (tuning key 19 = 1)
    int packed;                                           // MFMA backward (round 5): the ds / P' workspace is ONE array of (bf16 ds | bf16 P') words at `pm`
    const float* gs2;                                     // fp16-gradient backward (round 6, amdnuwa_sparse3dna_bwd_f16): device {S, 1 / S}; q / k / v / dO hold fp16 values,
                                                          // dO = fp16(S dO), dq / dk / dv leave as fp16(S gradient), the workspace words are (fp16 S ds | fp16 P'), dW_th leaves times 1 / S
};
// a 16-bit element of an operand array as fp32: bf16, or (F16) fp16
template <bool F16> __device__ __forceinline__ float ld16_t(bf16_t v) { return F16 ? (float)__builtin_bit_cast(_Float16, v) : bf2f(v); }

constexpr float NEG_MAX = -3.4028234663852886e38f;

// Workgroups are handed to the 8 XCDs round-robin by linear id; every XCD has its own L2.  Consecutive query rows share
// almost all of their key / value rows, so the logical row id is remapped to give each XCD one CONTIGUOUS range of rows
// (whole samples): its L2 then holds the few frames in flight instead of an eighth of everything (bijective for any grid).
__device__ __forceinline__ int xcd_row_id() {
    const int nb = gridDim.x, id = blockIdx.x, per = nb >> 3, rem = nb & 7, x = id & 7, k = id >> 3;
    return x * per + (x < rem ? x : rem) + k;
}


// hs = element stride between the 8-element halves of a chunk (8 = contiguous; the LDS stage keeps the two
// 16-byte halves of every chunk in separate regions so that ds_read_b128 lanes are 16 bytes apart: no conflicts)
template <int CH>
__device__ __forceinline__ void load_chunk(const bf16_t* hi, const bf16_t* lo, float* f, int hs = 8) {
#pragma unroll
    for (int v8 = 0; v8 < CH / 8; ++v8) {
        const uint4 u = *reinterpret_cast<const uint4*>(hi + v8 * hs);
        f[v8 * 8 + 0] = lo_f(u.x); f[v8 * 8 + 1] = hi_f(u.x); f[v8 * 8 + 2] = lo_f(u.y); f[v8 * 8 + 3] = hi_f(u.y);
        f[v8 * 8 + 4] = lo_f(u.z); f[v8 * 8 + 5] = hi_f(u.z); f[v8 * 8 + 6] = lo_f(u.w); f[v8 * 8 + 7] = hi_f(u.w);
        if (lo) {
            const uint4 l = *reinterpret_cast<const uint4*>(lo + v8 * hs);
            f[v8 * 8 + 0] += lo_f(l.x); f[v8 * 8 + 1] += hi_f(l.x); f[v8 * 8 + 2] += lo_f(l.y); f[v8 * 8 + 3] += hi_f(l.y);
            f[v8 * 8 + 4] += lo_f(l.z); f[v8 * 8 + 5] += hi_f(l.z); f[v8 * 8 + 6] += lo_f(l.w); f[v8 * 8 + 7] += hi_f(l.w);
        }
    }
}
template <int CH>
__device__ __forceinline__ void store_chunk(bf16_t* hi, bf16_t* lo, const float* f) {
#pragma unroll
    for (int v8 = 0; v8 < CH / 8; ++v8) {
        bf16_t h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f2bf_hilo(f[v8 * 8 + e], h[e], l[e]);
        *reinterpret_cast<uint4*>(hi + v8 * 8) = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
        if (lo) *reinterpret_cast<uint4*>(lo + v8 * 8) = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
    }
}
// reductions over the 4 lanes of a (query, head) group with DPP quad permutes (no LDS traffic)
__device__ __forceinline__ float dpp_quad(float v, int ctrl_b1) {
    const int i = __builtin_bit_cast(int, v);
    const int r = ctrl_b1 ? __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true)     // quad_perm [1,0,3,2]
                          : __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_quad(v, 1);
    v += dpp_quad(v, 0);
    return v;
}
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, dpp_quad(v, 1));
    v = fmaxf(v, dpp_quad(v, 0));
    return v;
}

// packed bf16 arithmetic (bf16 operand mode): v_dot2_f32_bf16 multiplies two bf16 pairs and accumulates in fp32
template <int CH>
__device__ __forceinline__ void load_pk(const bf16_t* p, uint32_t* pk, int hs = 8) {
#pragma unroll
    for (int v8 = 0; v8 < CH / 8; ++v8) {
        const uint4 u = *reinterpret_cast<const uint4*>(p + v8 * hs);
        pk[v8 * 4 + 0] = u.x; pk[v8 * 4 + 1] = u.y; pk[v8 * 4 + 2] = u.z; pk[v8 * 4 + 3] = u.w;
    }
}
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
template <int CH>
__device__ __forceinline__ float dot_pk(const uint32_t* a, const uint32_t* b) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < CH / 2; i += 2) { s0 = dot2(a[i], b[i], s0); s1 = dot2(a[i + 1], b[i + 1], s1); }
    return s0 + s1;
}
// acc[e] += coef * v[e] with coef rounded to bf16: one dot2 per element, no unpacking
template <int CH>
__device__ __forceinline__ void axpy_pk(float* acc, float coef, const uint32_t* v) {
    const uint32_t clo = f2bf(coef), chi = clo << 16;
#pragma unroll
    for (int i = 0; i < CH / 2; ++i) {
        acc[2 * i] = dot2(v[i], clo, acc[2 * i]);
        acc[2 * i + 1] = dot2(v[i], chi, acc[2 * i + 1]);
    }
}

// Sweep over all causal taps of one query row.  One key/value grid row (all heads) is staged in LDS per
// round; the NEXT valid row is fetched into registers while the current one is consumed.  The tap planes
// (ta, tb) are enumerated incrementally (fr += df, yr += dh): no integer division on the scalar unit.
// LDS image of a staged row: half v8 of the chunk of (w, h, c) lives at v8*HS + ((w*NH + h)*4 + c)*8
// (HS = W*NH*32 elements), so consecutive lanes read consecutive 16-byte words (conflict-free ds_read_b128).
// fn(j, hi, lo, hs) is called for every valid tap slot j >= 1 of this thread's (query, head).
constexpr int KHMAX = 3;   // (kept for the launch code; slab staging was measured slower and removed)

struct PlaneIt { int ta, tb, fr, yr; };

template <int CH, typename Fn>
__device__ __forceinline__ void sweep_taps(const S3Args& a, const bf16_t* src, const bf16_t* srcl, int b, int f, int y, int w,
                                           int h, int c, bool act, bool qvalid, bf16_t* st_hi, Fn&& fn) {
    const int HS = a.W * a.NH * 32;                      // elements per half image
    bf16_t* st_lo = st_hi + (CH / 8) * HS;
    const int myslot = ((w * a.NH + h) * 4 + c) * 8;
    const int yr0 = y - a.oh * a.dh;
    uint4 rh[CH / 8], rl[CH / 8];
    auto seek = [&](PlaneIt& p) {                        // advance to a valid plane (or ta == kf)
        while (p.ta < a.kf && !(p.fr >= 0 && p.yr >= 0 && p.fr < a.FK && p.yr < a.H)) {
            ++p.tb; p.yr += a.dh;
            if (p.tb == a.kh) { p.tb = 0; p.yr = yr0; ++p.ta; p.fr += a.df; }
        }
    };
    auto step = [&](PlaneIt& p) {
        ++p.tb; p.yr += a.dh;
        if (p.tb == a.kh) { p.tb = 0; p.yr = yr0; ++p.ta; p.fr += a.df; }
        seek(p);
    };
    auto fetch = [&](const PlaneIt& p) {
        const int pos = (p.fr * a.H + p.yr) * a.W + w;
        const bool ok = act && (a.kvoff + pos) < a.kvrows;
        const size_t gi = ((size_t)b * a.kvrows + a.kvoff + pos) * a.ldk + h * (CH * 4) + c * CH;
#pragma unroll
        for (int v8 = 0; v8 < CH / 8; ++v8) {
            rh[v8] = ok ? *reinterpret_cast<const uint4*>(src + gi + v8 * 8) : make_uint4(0, 0, 0, 0);
            if (srcl) rl[v8] = ok ? *reinterpret_cast<const uint4*>(srcl + gi + v8 * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    PlaneIt cur{0, 0, (a.xmode ? 0 : f) - a.of * a.df, yr0};
    seek(cur);
    if (cur.ta < a.kf) fetch(cur);
    while (cur.ta < a.kf) {
        __syncthreads();                                 // previous row fully consumed
        if (act) {
#pragma unroll
            for (int v8 = 0; v8 < CH / 8; ++v8) {
                *reinterpret_cast<uint4*>(st_hi + v8 * HS + myslot) = rh[v8];
                if (srcl) *reinterpret_cast<uint4*>(st_lo + v8 * HS + myslot) = rl[v8];
            }
        }
        __syncthreads();
        PlaneIt nxt = cur;
        step(nxt);
        if (nxt.ta < a.kf) fetch(nxt);                   // in flight during the compute below
        if (qvalid) {
            const int jb = 1 + (cur.ta * a.kh + cur.tb) * a.kw;
            int wr = w - a.ow * a.dw;
            const uint8_t* mrow = a.kmask ? a.kmask + (size_t)b * a.kvrows + a.kvoff + (cur.fr * a.H + cur.yr) * a.W : nullptr;
            for (int tc = 0; tc < a.kw; ++tc, wr += a.dw) {
                if (wr < 0 || wr >= a.W) continue;
                if (mrow && !mrow[wr]) continue;          // masked key: its slot keeps the mask value (P = 0) in every sweep
                const int slot = ((wr * a.NH + h) * 4 + c) * 8;
                fn(jb + tc, st_hi + slot, srcl ? st_lo + slot : nullptr, HS);
            }
        }
        cur = nxt;
    }
}

// scores + softmax for one query row: fills SP[(w*J + j)*NH + h] with P (fp32).  Shared by fwd and bwd_q.
template <int DH, bool LO>
__device__ __forceinline__ void scores_softmax(const S3Args& a, int b, int f, int y, int w, int h, int c, bool act, bool qvalid,
                                               const float* qf, const uint32_t* qp, float* SP, bf16_t* st_hi, int J) {
    constexpr int CH = DH / 4;
    const int t = threadIdx.x, nt = blockDim.x;
    for (int e = t; e < a.W * J * a.NH; e += nt) SP[e] = NEG_MAX;
    __syncthreads();
    // <bos> key: slot j = 0
    auto qk = [&](const bf16_t* khi, const bf16_t* klo, int hs) {
        float s = 0.f;
        if (LO) {
            float kf_[CH];
            load_chunk<CH>(khi, klo, kf_, hs);
#pragma unroll
            for (int e = 0; e < CH; ++e) s += qf[e] * kf_[e];
        } else {
            uint32_t kp[CH / 2];
            load_pk<CH>(khi, kp, hs);
            s = dot_pk<CH>(qp, kp);
        }
        return quad_sum(s);
    };
    if (qvalid) {
        const size_t g = (size_t)b * a.k0_bs + h * DH + c * CH;
        const float s = qk(a.k0 + g, a.k0l ? a.k0l + g : nullptr, 8);
        if (c == 0) SP[(w * J + 0) * a.NH + h] = s * a.scale + (a.bias ? a.bias[h] : 0.f);
    }
    sweep_taps<CH>(a, a.k, a.kl, b, f, y, w, h, c, act, qvalid, st_hi, [&](int j, const bf16_t* khi, const bf16_t* klo, int hs) {
        const float s = qk(khi, klo, hs);
        if (c == 0) SP[(w * J + j) * a.NH + h] = s * a.scale + (a.bias ? a.bias[j * a.NH + h] : 0.f);
    });
    __syncthreads();
    // fp32 softmax over the J slots of each (w, h): the 4 lanes of the group split j
    if (act) {
        float m = NEG_MAX;
        for (int j = c; j < J; j += 4) m = fmaxf(m, SP[(w * J + j) * a.NH + h]);
        m = quad_max(m);
        float s = 0.f;
        for (int j = c; j < J; j += 4) s += expf(SP[(w * J + j) * a.NH + h] - m);
        s = quad_sum(s);
        const float inv = 1.f / s;
        for (int j = c; j < J; j += 4) {
            const int idx = (w * J + j) * a.NH + h;
            SP[idx] = expf(SP[idx] - m) * inv;
        }
    }
    __syncthreads();
}

template <int DH, bool LO>
__global__ __launch_bounds__(512, 4) void s3_fwd_kernel(S3Args a) {
    constexpr int CH = DH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int J = a.kf * a.kh * a.kw + 1;
    const int stage_elems = a.W * a.NH * DH;
    bf16_t* st_hi = reinterpret_cast<bf16_t*>(smem);
    float* SP = reinterpret_cast<float*>(smem + (size_t)stage_elems * (a.kl ? 4 : 2));
    __shared__ float wsh[64];
    const int t = threadIdx.x, c = t & 3, wh = t >> 2, h = wh % a.NH, w = wh / a.NH;
    const bool act = w < a.W;
    const int rows = a.F * a.H;
    const int bid = xcd_row_id();
    const int b = bid / rows, ry = bid % rows, f = ry / a.H, y = ry % a.H;
    const int i = 1 + ry * a.W + w;
    const bool qvalid = act && i < a.ntok;
    if (t < a.NH * a.NH) wsh[t] = a.wth[t];
    // <bos> output row = its own value (np.py:499, 608)
    if (ry == 0 && !a.xmode) {
        const int inner = a.NH * DH;
        for (int e = t; e < inner; e += blockDim.x) {
            const size_t gi = ((size_t)b * a.ntok) * a.ld + e, go = ((size_t)b * a.ntok) * a.ldo + e;
            a.o[go] = a.v[gi];
            if (a.ol) a.ol[go] = a.vl ? a.vl[gi] : (bf16_t)0;
        }
    }
    if (ry * a.W + 1 >= a.ntok) return;   // whole row beyond the sequence (uniform)
    float qf[CH];
    uint32_t qp[CH / 2];
    if (qvalid) {
        const size_t g = ((size_t)b * a.ntok + i) * a.ld + h * DH + c * CH;
        if (LO) load_chunk<CH>(a.q + g, a.ql ? a.ql + g : nullptr, qf); else load_pk<CH>(a.q + g, qp);
    }
    scores_softmax<DH, LO>(a, b, f, y, w, h, c, act, qvalid, qf, qp, SP, st_hi, J);
    // talking heads: P'[g] = sum_h Wth[g][h] P[h] per (w, j), in place
    for (int item = t; item < a.W * J; item += blockDim.x) {
        float pv[8], out[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) pv[hh] = hh < a.NH ? SP[item * a.NH + hh] : 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float s = 0.f;
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) s += (g < a.NH && hh < a.NH ? wsh[g * a.NH + hh] : 0.f) * pv[hh];
            out[g] = s;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) if (g < a.NH) SP[item * a.NH + g] = out[g];
    }
    __syncthreads();
    // P'.V
    float of[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) of[e] = 0.f;
    auto pv = [&](float pj, const bf16_t* vhi, const bf16_t* vlo, int hs) {
        if (LO) {
            float vf[CH];
            load_chunk<CH>(vhi, vlo, vf, hs);
#pragma unroll
            for (int e = 0; e < CH; ++e) of[e] += pj * vf[e];
        } else {
            uint32_t vp[CH / 2];
            load_pk<CH>(vhi, vp, hs);
            axpy_pk<CH>(of, pj, vp);
        }
    };
    if (qvalid) {
        const size_t g = (size_t)b * a.k0_bs + h * DH + c * CH;
        pv(SP[(w * J + 0) * a.NH + h], a.v0 + g, a.v0l ? a.v0l + g : nullptr, 8);
    }
    sweep_taps<CH>(a, a.v, a.vl, b, f, y, w, h, c, act, qvalid, st_hi, [&](int j, const bf16_t* vhi, const bf16_t* vlo, int hs) {
        pv(SP[(w * J + j) * a.NH + h], vhi, vlo, hs);
    });
    if (qvalid) {
        const size_t g = ((size_t)b * a.ntok + i) * a.ldo + h * DH + c * CH;
        store_chunk<CH>(a.o + g, a.ol ? a.ol + g : nullptr, of);
    }
}

template <int DH, bool LO>
__global__ __launch_bounds__(512, 4) void s3_bwd_q_kernel(S3Args a) {
    constexpr int CH = DH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int J = a.kf * a.kh * a.kw + 1;
    const int stage_elems = a.W * a.NH * DH;
    const int nsp = a.W * J * a.NH;
    bf16_t* st_hi = reinterpret_cast<bf16_t*>(smem);
    float* SP = reinterpret_cast<float*>(smem + (size_t)stage_elems * (a.kl ? 4 : 2));   // P, later unchanged
    float* DP = SP + nsp;                                                   // dP' -> dP -> ds
    float* RED = DP + nsp;                                                  // [8][64] dW_th partials / [W][inner] bos partials
    __shared__ float wsh[64];
    const int t = threadIdx.x, c = t & 3, wh = t >> 2, h = wh % a.NH, w = wh / a.NH;
    const bool act = w < a.W;
    const int rows = a.F * a.H, inner = a.NH * DH;
    const int bid = xcd_row_id();
    const int b = bid / rows, ry = bid % rows, f = ry / a.H, y = ry % a.H;
    const int i = 1 + ry * a.W + w;
    const bool qvalid = act && i < a.ntok;
    const int nq = a.ntok - 1;
    if (t < a.NH * a.NH) wsh[t] = a.wth[t];
    float* pth = a.part_th + (size_t)bid * a.NH * a.NH;
    float* pk0 = a.part_k0 + (size_t)bid * inner;
    float* pv0 = a.part_v0 + (size_t)bid * inner;
    if (ry == 0 && !a.xmode) {   // dq of the <bos> row is zero (its query is never used)
        for (int e = t; e < inner; e += blockDim.x) {
            const size_t go = ((size_t)b * a.ntok) * a.ldd + e;
            a.dq[go] = 0;
            if (a.dql) a.dql[go] = 0;
        }
    }
    if (ry * a.W + 1 >= a.ntok) {   // row beyond the sequence: contributes nothing
        for (int e = t; e < a.NH * a.NH; e += blockDim.x) pth[e] = 0.f;
        for (int e = t; e < inner; e += blockDim.x) { pk0[e] = 0.f; pv0[e] = 0.f; }
        return;
    }
    float qf[CH], dof[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { qf[e] = 0.f; dof[e] = 0.f; }
    uint32_t qp[CH / 2], dop[CH / 2];            // packed bf16 q / dO for the dot2 path (bf16 operand mode)
#pragma unroll
    for (int e = 0; e < CH / 2; ++e) { qp[e] = 0; dop[e] = 0; }
    if (qvalid) {
        const size_t g = ((size_t)b * a.ntok + i) * a.ld + h * DH + c * CH;
        const size_t gd = ((size_t)b * a.ntok + i) * a.lddo + h * DH + c * CH;
        if (LO) {
            load_chunk<CH>(a.q + g, a.ql ? a.ql + g : nullptr, qf);
            load_chunk<CH>(a.dO + gd, a.dOl ? a.dOl + gd : nullptr, dof);
        } else {
            load_pk<CH>(a.q + g, qp);
            load_pk<CH>(a.dO + gd, dop);
        }
    }
    scores_softmax<DH, LO>(a, b, f, y, w, h, c, act, qvalid, qf, qp, SP, st_hi, J);
    // P' = mix(P) -> global (needed by bwd_kv); P stays in SP
    for (int item = t; item < a.W * J; item += blockDim.x) {
        const int wq = item / J, j = item % J;
        const int iq = 1 + ry * a.W + wq;
        float pv[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) pv[hh] = hh < a.NH ? SP[item * a.NH + hh] : 0.f;
        if (iq < a.ntok) {
            float* dst = a.pm + (((size_t)b * nq + (iq - 1)) * J + j) * a.NH;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float s = 0.f;
#pragma unroll
                for (int hh = 0; hh < 8; ++hh) s += (g < a.NH && hh < a.NH ? wsh[g * a.NH + hh] : 0.f) * pv[hh];
                if (g < a.NH) dst[g] = s;
            }
        }
    }
    for (int e = t; e < nsp; e += blockDim.x) DP[e] = 0.f;
    __syncthreads();
    // dP'[w][j][g] = dO[w][g] . v_j[g]
    auto dov = [&](const bf16_t* vhi, const bf16_t* vlo, int hs) {
        float s = 0.f;
        if (LO) {
            float vf[CH];
            load_chunk<CH>(vhi, vlo, vf, hs);
#pragma unroll
            for (int e = 0; e < CH; ++e) s += dof[e] * vf[e];
        } else {
            uint32_t vp[CH / 2];
            load_pk<CH>(vhi, vp, hs);
            s = dot_pk<CH>(dop, vp);
        }
        return quad_sum(s);
    };
    if (qvalid) {
        const size_t g = (size_t)b * a.k0_bs + h * DH + c * CH;
        const float s = dov(a.v0 + g, a.v0l ? a.v0l + g : nullptr, 8);
        if (c == 0) DP[(w * J + 0) * a.NH + h] = s;
    }
    sweep_taps<CH>(a, a.v, a.vl, b, f, y, w, h, c, act, qvalid, st_hi, [&](int j, const bf16_t* vhi, const bf16_t* vlo, int hs) {
        const float s = dov(vhi, vlo, hs);
        if (c == 0) DP[(w * J + j) * a.NH + h] = s;
    });
    __syncthreads();
    // dW_th[g][h] partial = sum_{w,j} dP'[g] * P[h]   (thread = (g,h) pair x 8 item groups)
    {
        const int pair = t & 63, grp = t >> 6, ng = blockDim.x >> 6;
        const int g = pair / a.NH, hh = pair % a.NH;
        float acc = 0.f;
        if (pair < a.NH * a.NH)
            for (int item = grp; item < a.W * J; item += ng) acc += DP[item * a.NH + g] * SP[item * a.NH + hh];
        RED[grp * 64 + pair] = acc;
        __syncthreads();
        if (t < a.NH * a.NH) {
            float s = 0.f;
            for (int k = 0; k < ng; ++k) s += RED[k * 64 + t];
            pth[t] = s;
        }
        __syncthreads();
    }
    // dP[h] = sum_g Wth[g][h] dP'[g]   (in place, item-local)
    for (int item = t; item < a.W * J; item += blockDim.x) {
        float dv_[8], out[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) dv_[g] = g < a.NH ? DP[item * a.NH + g] : 0.f;
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += (g < a.NH && hh < a.NH ? wsh[g * a.NH + hh] : 0.f) * dv_[g];
            out[hh] = s;
        }
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) if (hh < a.NH) DP[item * a.NH + hh] = out[hh];
    }
    __syncthreads();
    // ds = P * (dP - sum_j P dP)
    if (act) {
        float d = 0.f;
        for (int j = c; j < J; j += 4) d += SP[(w * J + j) * a.NH + h] * DP[(w * J + j) * a.NH + h];
        d = quad_sum(d);
        for (int j = c; j < J; j += 4) {
            const int idx = (w * J + j) * a.NH + h;
            const float dsv = SP[idx] * (DP[idx] - d);
            DP[idx] = dsv;
            if (qvalid) a.ds[(((size_t)b * nq + (i - 1)) * J + j) * a.NH + h] = dsv;
        }
    }
    __syncthreads();
    // dq = scale * sum_j ds_j k_j ;  <bos> partials: dk0 += scale*ds_0*q, dv0 += P'_0*dO
    float dqf[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) dqf[e] = 0.f;
    float k0c[CH], v0c[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { k0c[e] = 0.f; v0c[e] = 0.f; }
    if (qvalid) {
        float kf_[CH];
        const size_t g = (size_t)b * a.k0_bs + h * DH + c * CH;
        load_chunk<CH>(a.k0 + g, a.k0l ? a.k0l + g : nullptr, kf_);
        const float d0 = DP[(w * J + 0) * a.NH + h];
        float pm0 = 0.f;                                        // P'[w][0][g = h] = sum_hh Wth[h][hh] P[w][0][hh]
        for (int hh = 0; hh < a.NH; ++hh) pm0 += wsh[h * a.NH + hh] * SP[(w * J + 0) * a.NH + hh];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const float qe = LO ? qf[e] : ((e & 1) ? hi_f(qp[e >> 1]) : lo_f(qp[e >> 1]));
            const float de = LO ? dof[e] : ((e & 1) ? hi_f(dop[e >> 1]) : lo_f(dop[e >> 1]));
            dqf[e] += d0 * kf_[e];
            k0c[e] = a.scale * d0 * qe;
            v0c[e] = pm0 * de;
        }
    }
    sweep_taps<CH>(a, a.k, a.kl, b, f, y, w, h, c, act, qvalid, st_hi, [&](int j, const bf16_t* khi, const bf16_t* klo, int hs) {
        const float dj = DP[(w * J + j) * a.NH + h];
        if (LO) {
            float kf_[CH];
            load_chunk<CH>(khi, klo, kf_, hs);
#pragma unroll
            for (int e = 0; e < CH; ++e) dqf[e] += dj * kf_[e];
        } else {
            uint32_t kp[CH / 2];
            load_pk<CH>(khi, kp, hs);
            axpy_pk<CH>(dqf, dj, kp);
        }
    });
    if (qvalid) {
#pragma unroll
        for (int e = 0; e < CH; ++e) dqf[e] *= a.scale;
        const size_t g = ((size_t)b * a.ntok + i) * a.ldd + h * DH + c * CH;
        store_chunk<CH>(a.dq + g, a.dql ? a.dql + g : nullptr, dqf);
    }
    // reduce the <bos> partials over the W queries of the row (fixed order), via LDS (reuses SP/DP space),
    // one of the two at a time: needs W*inner floats <= 2*nsp (checked on the host)
    float* RK = SP;                       // [W][inner]
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
        if (act) {
#pragma unroll
            for (int e = 0; e < CH; ++e) RK[w * inner + h * DH + c * CH + e] = which ? v0c[e] : k0c[e];
        }
        __syncthreads();
        for (int e = t; e < inner; e += blockDim.x) {
            float sk = 0.f;
            for (int ww = 0; ww < a.W; ++ww) sk += RK[ww * inner + e];
            (which ? pv0 : pk0)[e] = sk;
        }
    }
}

template <int DH, bool LO>
__global__ __launch_bounds__(512, 4) void s3_bwd_kv_kernel(S3Args a) {
    constexpr int CH = DH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int J = a.kf * a.kh * a.kw + 1;
    const int stage_elems = a.W * a.NH * DH;
    bf16_t* sq_hi = reinterpret_cast<bf16_t*>(smem);
    bf16_t* sq_lo = sq_hi + stage_elems;
    bf16_t* sd_hi = sq_lo + stage_elems;
    bf16_t* sd_lo = sd_hi + stage_elems;
    const int t = threadIdx.x, c = t & 3, wh = t >> 2, h = wh % a.NH, w = wh / a.NH;
    const bool act = w < a.W;
    const int rows = a.FK * a.H;                     // key rows of the grid (the launch covers B * FK * H workgroups)
    const int bid = xcd_row_id();
    const int b = bid / rows, ry = bid % rows, f = ry / a.H, y = ry % a.H;
    const int ik = a.kvoff + ry * a.W + w;           // key row index inside the sample
    const bool kvalid = act && ik < a.kvrows;
    const int nq = a.ntok - 1;
    if (ry * a.W + a.kvoff >= a.kvrows) return;
    float dkf[CH], dvf[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { dkf[e] = 0.f; dvf[e] = 0.f; }
    // planes t = ta*kh + tb; the attending query row of plane t is (f + (of-ta)df, y + (oh-tb)dh)  [of = kf-1 when causal]
    // (xmode: key frame f is tap f of EVERY query frame, so the sweep runs over the query frames instead of the frame taps)
    const int nplanes = (a.xmode ? a.F : a.kf) * a.kh;
    auto plane = [&](int t, int& fq, int& yq) {
        const int ta = t / a.kh, tb = t - ta * a.kh;
        fq = a.xmode ? ta : f + (a.of - ta) * a.df;
        yq = y + (a.oh - tb) * a.dh;
        return fq >= 0 && yq >= 0 && fq < a.F && yq < a.H && (fq * a.H + yq) * a.W + 1 < a.ntok;
    };
    auto slot_plane = [&](int t) { return a.xmode ? f * a.kh + (t % a.kh) : t; };      // tap-plane index inside the J slots
    auto next_plane = [&](int t) { int fq, yq; while (t < nplanes && !plane(t, fq, yq)) ++t; return t; };
    uint4 rq[CH / 8], rd[CH / 8], rql[CH / 8], rdl[CH / 8];
    constexpr int KWMAX = 3;                        // taps prefetched one plane ahead; wider kernels load the rest directly
    float nds[KWMAX], npm[KWMAX];                   // ds / P' of the NEXT plane's taps (this thread's key, head): loaded one plane ahead
    auto fetch = [&](int t) {                       // q row and dO row of the attending query row -> registers
        int fq, yq;
        plane(t, fq, yq);
#pragma unroll
        for (int tc = 0; tc < KWMAX; ++tc) {
            nds[tc] = npm[tc] = 0.f;
            const int wq = w + (a.ow - tc) * a.dw;
            const int pqn = (fq * a.H + yq) * a.W + wq;
            if (tc < a.kw && kvalid && wq >= 0 && wq < a.W && 1 + pqn < a.ntok) {
                const size_t ci = (((size_t)b * nq + pqn) * J + 1 + slot_plane(t) * a.kw + tc) * a.NH + h;
                nds[tc] = a.ds[ci]; npm[tc] = a.pm[ci];
            }
        }
        const int pq = (fq * a.H + yq) * a.W + w;
        const bool ok = act && (1 + pq) < a.ntok;
        const size_t gq = ((size_t)b * a.ntok + 1 + pq) * a.ld + h * DH + c * CH;
        const size_t gd = ((size_t)b * a.ntok + 1 + pq) * a.lddo + h * DH + c * CH;
#pragma unroll
        for (int v8 = 0; v8 < CH / 8; ++v8) {
            rq[v8] = ok ? *reinterpret_cast<const uint4*>(a.q + gq + v8 * 8) : make_uint4(0, 0, 0, 0);
            rd[v8] = ok ? *reinterpret_cast<const uint4*>(a.dO + gd + v8 * 8) : make_uint4(0, 0, 0, 0);
            if (a.ql) rql[v8] = ok ? *reinterpret_cast<const uint4*>(a.ql + gq + v8 * 8) : make_uint4(0, 0, 0, 0);
            if (a.dOl) rdl[v8] = ok ? *reinterpret_cast<const uint4*>(a.dOl + gd + v8 * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    const int HS = a.W * a.NH * 32;               // half-split LDS image (see sweep_taps): conflict-free b128 reads
    const int myslot = ((w * a.NH + h) * 4 + c) * 8;
    int tp = next_plane(0);
    if (tp < nplanes) fetch(tp);
    while (tp < nplanes) {
        __syncthreads();
        if (act) {
#pragma unroll
            for (int v8 = 0; v8 < CH / 8; ++v8) {
                *reinterpret_cast<uint4*>(sq_hi + myslot + v8 * HS) = rq[v8];
                *reinterpret_cast<uint4*>(sd_hi + myslot + v8 * HS) = rd[v8];
                if (a.ql) *reinterpret_cast<uint4*>(sq_lo + myslot + v8 * HS) = rql[v8];
                if (a.dOl) *reinterpret_cast<uint4*>(sd_lo + myslot + v8 * HS) = rdl[v8];
            }
        }
        __syncthreads();
        float cds[KWMAX], cpm[KWMAX];
#pragma unroll
        for (int tc = 0; tc < KWMAX; ++tc) { cds[tc] = nds[tc]; cpm[tc] = npm[tc]; }
        const int tn = next_plane(tp + 1);
        if (tn < nplanes) fetch(tn);                // in flight during the FMAs below
        if (kvalid) {
            int fq, yq;
            plane(tp, fq, yq);
            for (int tc = 0; tc < a.kw; ++tc) {
                const int wq = w + (a.ow - tc) * a.dw;
                if (wq < 0 || wq >= a.W) continue;
                const int pq = (fq * a.H + yq) * a.W + wq;
                if (1 + pq >= a.ntok) continue;
                float dsv, pmv;
                if (tc < KWMAX) { dsv = tc == 0 ? cds[0] : (tc == 1 ? cds[1] : cds[2]); pmv = tc == 0 ? cpm[0] : (tc == 1 ? cpm[1] : cpm[2]); }
                else {
                    const size_t ci = (((size_t)b * nq + pq) * J + 1 + slot_plane(tp) * a.kw + tc) * a.NH + h;
                    dsv = a.ds[ci]; pmv = a.pm[ci];
                }
                const int slot = ((wq * a.NH + h) * 4 + c) * 8;
                if (LO) {
                    float qq[CH], dd[CH];
                    load_chunk<CH>(sq_hi + slot, a.ql ? sq_lo + slot : nullptr, qq, HS);
                    load_chunk<CH>(sd_hi + slot, a.dOl ? sd_lo + slot : nullptr, dd, HS);
#pragma unroll
                    for (int e = 0; e < CH; ++e) { dkf[e] += dsv * qq[e]; dvf[e] += pmv * dd[e]; }
                } else {
                    uint32_t qk2[CH / 2], dk2[CH / 2];
                    load_pk<CH>(sq_hi + slot, qk2, HS);
                    load_pk<CH>(sd_hi + slot, dk2, HS);
                    axpy_pk<CH>(dkf, dsv, qk2);
                    axpy_pk<CH>(dvf, pmv, dk2);
                }
            }
        }
        tp = tn;
    }
    if (kvalid) {
#pragma unroll
        for (int e = 0; e < CH; ++e) dkf[e] *= a.scale;
        const size_t g = ((size_t)b * a.kvrows + ik) * a.lddk + h * DH + c * CH;
        store_chunk<CH>(a.dk + g, a.dkl ? a.dkl + g : nullptr, dkf);
        store_chunk<CH>(a.dv + g, a.dvl ? a.dvl + g : nullptr, dvf);
    }
}

// fixed-order reductions: dW_th (+= over all workgroups); per sample dk[bos], dv[bos] (+ dO[bos])
// 1024 threads = 64 columns x 16 row groups (coalesced reads, fixed combine order).
// grid = B * ceil(inner/64) blocks for the <bos> k/v rows + ceil(NH*NH/16) blocks for dW_th.
__global__ __launch_bounds__(1024) void s3_bwd_fin_kernel(S3Args a, int DH) {
    __shared__ float red[2][16][64];
    const int rows = a.F * a.H, inner = a.NH * DH;
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int nchunk = (inner + 63) / 64;
    if ((int)blockIdx.x >= a.B * nchunk) {
        // dW_th: the last ceil(NH*NH / 16) blocks, each 16 columns x 64 row groups over the B*F*H workgroup partials
        // (one block summing all of them serially was a 130 us tail on the backward)
        __shared__ float redw[64][17];
        const int nn = a.NH * a.NH;
        const int col = (blockIdx.x - a.B * nchunk) * 16 + (threadIdx.x & 15), rg64 = threadIdx.x >> 4;
        float s = 0.f;
        if (col < nn) {
            // eight partials in flight per thread, added in their index order (one dependent 4-byte load per iteration made this
            // block the 80-us critical path of the kernel); indices past the end re-read row 0 and add zero
            const int n = a.B * rows;
            for (int k = rg64; k < n; k += 64 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ku = k + 64 * u;
                    const float t = a.part_th[(size_t)(ku < n ? ku : 0) * nn + col];
                    v[u] = ku < n ? t : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        }
        redw[rg64][threadIdx.x & 15] = s;
        __syncthreads();
        if (rg64 == 0 && col < nn) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 64; ++r) t += redw[r][threadIdx.x & 15];
            t *= f16_gs_inv(a.gs2);                                   // (fp16-gradient form: the partials carry the factor S)
            a.dwth[col] = a.accumulate ? a.dwth[col] + t : t;
        }
        return;
    }
    if (a.xmode) {
        // SparseCross2DNA: slot 0 is ONE learned null key / value for every sample -> sum the partials of all B * rows workgroups
        // (blocks 0 .. nchunk-1; the remaining per-sample blocks have nothing to do)
        if ((int)blockIdx.x >= nchunk) return;
        const int e = blockIdx.x * 64 + lane;
        float sk = 0.f, sv = 0.f;
        if (e < inner)
            for (int r = rg; r < a.B * rows; r += 16) {
                sk += a.part_k0[(size_t)r * inner + e];
                sv += a.part_v0[(size_t)r * inner + e];
            }
        red[0][rg][lane] = sk;
        red[1][rg][lane] = sv;
        __syncthreads();
        if (rg == 0 && e < inner) {
            float tk = 0.f, tv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { tk += red[0][r][lane]; tv += red[1][r][lane]; }
            a.dnull_k[e] = tk;
            a.dnull_v[e] = tv;
        }
        return;
    }
    const int b = blockIdx.x / nchunk, e = (blockIdx.x % nchunk) * 64 + lane;
    float sk = 0.f, sv = 0.f;
    if (e < inner)
        for (int r = rg; r < rows; r += 16 * 4) {                // four row partials in flight per thread, added in order
            float vk[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + 16 * u;
                const size_t o = ((size_t)b * rows + (ru < rows ? ru : 0)) * inner + e;
                const float tk = a.part_k0[o], tv = a.part_v0[o];
                vk[u] = ru < rows ? tk : 0.f; vv[u] = ru < rows ? tv : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sk += vk[u]; sv += vv[u]; }
        }
    red[0][rg][lane] = sk;
    red[1][rg][lane] = sv;
    __syncthreads();
    if (rg == 0 && e < inner) {
        float tk = 0.f, tv = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { tk += red[0][r][lane]; tv += red[1][r][lane]; }
        const size_t gi = ((size_t)b * a.ntok) * a.lddo + e, go = ((size_t)b * a.ntok) * a.ldd + e;
        if (a.gs2) {                                                  // fp16-gradient form: fp16 in, fp16 out (still times S)
            tv += ld16_t<true>(a.dO[gi]);
            float amax = fmaxf(fabsf(tk), fabsf(tv));
            a.dk[go] = f2h_sat(tk); a.dv[go] = f2h_sat(tv);
            f16_sat_commit(amax);
            return;
        }
        tv += bf2f(a.dO[gi]) + (a.dOl ? bf2f(a.dOl[gi]) : 0.f);
        bf16_t hh, ll;
        f2bf_hilo(tk, hh, ll); a.dk[go] = hh; if (a.dkl) a.dkl[go] = ll;
        f2bf_hilo(tv, hh, ll); a.dv[go] = hh; if (a.dvl) a.dvl[go] = ll;
    }
}

int check_geom(const amdnuwa_s3_geom* g) {
    if (!g) return AMDNUWA_ERR_ARG;
    if (g->dim_head != 32 && g->dim_head != 64) return AMDNUWA_ERR_UNSUPPORTED;
    if (g->heads < 1 || g->heads > 8 || g->W < 1 || g->W * g->heads * 4 > 512) return AMDNUWA_ERR_UNSUPPORTED;
    if (g->kf < 1 || g->kh < 1 || g->kw < 1 || g->df < 1 || g->dh < 1 || g->dw < 1) return AMDNUWA_ERR_ARG;
    if (g->ntok < 1 || g->ntok - 1 > g->F * g->H * g->W) return AMDNUWA_ERR_ARG;
    return AMDNUWA_OK;
}
// Dynamic LDS of the window kernels grows with the window: J = kf*kh*kw + 1 key slots per query.  The largest request over the
// forward, the query-side and the key-side backward in the given operand form (lo = hi + lo pairs); the MFMA band kernels of
// the causal decoder shapes need less.  A launch above the CU's 160 KiB would fail inside hipLaunchKernel -- typically in the
// backward, after the forward succeeded -- so the entry points refuse the geometry up front (AMDNUWA_ERR_UNSUPPORTED) and
// amdnuwa_s3_supported() lets the host route such windows to its PyTorch-op formulation.
constexpr size_t S3_LDS_MAX = 160 * 1024 - 2048;        // leaves room for the kernels' static __shared__ arrays
size_t s3_lds_need(const amdnuwa_s3_geom* g, bool lo) {
    const size_t J = (size_t)g->kf * g->kh * g->kw + 1, inner = (size_t)g->heads * g->dim_head, W = (size_t)g->W;
    const size_t stage = W * inner * (lo ? 4 : 2), nsp = W * J * g->heads;
    const size_t fwd = stage + nsp * 4;
    size_t spdp = 2 * nsp;
    if (spdp < W * inner) spdp = W * inner;
    const size_t bq = stage + (spdp + 8 * 64) * 4;
    const size_t bkv = W * inner * 8;
    size_t m = fwd > bq ? fwd : bq;
    return m > bkv ? m : bkv;
}
// ---------------------------------------------------------------------------------------------
// MFMA forward (fast bf16 mode, W == 16, 8 heads x 64): one workgroup = one query row of the grid, wave h = head h.
//   phase 1  scores: for every causal tap plane the 16 keys of that grid row go STRAIGHT from global memory into the A operand
//            (S^T = K . Q^T, 2 x v_mfma_f32_16x16x32_bf16 per plane); a lane then owns 4 keys of one query and drops the
//            entries that lie on the kw tap diagonals into the compact score table SP[w][j][h] in LDS.  No LDS staging and no
//            workgroup barrier inside the sweep: the eight waves run free.
//   phase 2  fp32 softmax over the J slots and the talking-heads mix, in place (as the VALU kernel does).
//   phase 3  O^T = V^T . P'^T: the value rows of TWO planes (32 keys) are staged in a wave-private 4 KiB LDS tile and read back
//            transposed (ds_read_b64_tr_b16) as the A operand; the banded P' of the lane's 8 key slots is rebuilt from SP.
// Instruction count per wave is ~3x below the dot2 kernel's and the only workgroup barriers are the two around phase 2.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
__device__ __forceinline__ int vt_off(int row, int gc) { return row * 128 + ((gc ^ (row & 7)) << 4); }
__device__ __forceinline__ bf16x8 vt_tr(const char* tile, int db, int c, int g4) {
    // lane (c, g4) gets tile[kb*16 + 4*g4 + j][db*16 + c], j = 0..3, kb = 0, 1
    const int col = db * 16 + ((c & 3) << 2);
    const int r0 = 4 * g4 + (c >> 2), r1 = 16 + r0;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + vt_off(r1, col >> 3) + ((col >> 2) & 1) * 8));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 ldg8(const bf16_t* p, bool ok) {
    return __builtin_bit_cast(bf16x8, ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0));
}

#ifndef S3M_PF
#define S3M_PF 3
#endif
#ifndef S3M_BRANCHFREE
#define S3M_BRANCHFREE 0          // 1 = row fetches of the single-row sweeps without a branch around the load.  Measured SLOWER again in round 4 (A/B of
                                  // two builds in one call, b = 128: forward +1.5..4 %, backward +2..3 %; profiles/r04e_s3_branchfree_ab.txt): off
#endif
constexpr int S3M_KW = 3;        // widest tap row the MFMA kernel handles
constexpr int S3M_PLANES = 64;   // kf * kh limit (plane list in LDS)

// per-workgroup state of one query row, shared by the MFMA forward and backward kernels
struct RowM {
    int nplanes, J, TS, lane, wave, c, g4, iq;  // lane = (c = MFMA column / query, g4 = k-slice / row quad); TS = table stride per query
    bool qok;                                   // query c of this row exists (token iq < ntok)
    size_t tok0;                                // first token row of the sample
    int tsel[4];                                // tap index linking query c to key 4*g4 + r (the same in every plane), or -1
    const int *pslot, *ptok;                    // valid planes: first key slot j, token row of key 0
    int vps, vpt;                               // ... and the same lists with plane p in LANE p of a register (read back with v_readlane: see mfma_band_scores_fast)
};
constexpr int S3M_NH = 8, S3M_DH = 64, S3M_W = 16;
// Workgroup order inside a sample (a.ymajor; tuning key 3 bit 1 restores the frame-major order).  The ~64 workgroups an XCD runs
// at a time should share key rows that fit its 4 MB L2 (the whole sample's K + V is 5.2 MB).  Frame-major order keeps 4 query frames
// + up to 8 earlier key frames x 16 rows in flight: more than the L2 holds, every key row comes from HBM / MALL ~3 times.  Rows whose
// y agree modulo the dilation dh attend to key rows of the SAME residue class (y - 2 dh, y - dh, y), so the order is: residue class
// (y mod dh), then y, then frame -- a class is (H / dh) x F rows whose key rows are (H / dh) x F x 32 KB = 5.2 MB / dh.
__device__ __forceinline__ void s3m_row_order(const S3Args& a, int r2, int& f, int& y) {
    if (!a.ymajor) { f = r2 / a.H; y = r2 % a.H; return; }
    if (a.dh > 1 && a.H % a.dh == 0) {
        const int per = (a.H / a.dh) * a.F, cl = r2 / per, ii = r2 % per;
        y = cl + (ii / a.F) * a.dh; f = ii % a.F;
        return;
    }
    y = r2 / a.F; f = r2 % a.F;
}
// The score tables SP / DP of the MFMA kernels: entry (query w, slot j, head h) at word  w * TS + j * NH + h  with the per-query
// stride TS = J * NH + 4.  The band view of the sweeps (mfma_band_scores*, mfma_band_apply) walks the QUERY index across the 16
// lanes of an MFMA column group at fixed (slot, head); with the dense stride J * NH (= 368 words at J = 46: 16 mod 32) those 16
// lanes fell on two banks -- 8-way conflicts on every table access of the score and apply passes, the LDS pipe of a CU busy with
// conflict cycles for most of the backward (r02 PMC: SQ_LDS_BANK_CONFLICT = 67 % of the busy cycles of s3_bwd_q_mfma).  A pad of 4
// words (372: 20 mod 32) spreads the 16 queries over 8 banks and keeps every item's 8 heads 16-byte aligned for the item loops'
// ds_read_b128 pairs (a pad of 1 word reached 16 banks but turned those into eight 8-way-conflicting ds_read_b32: measured slower).
constexpr int S3M_PAD = 4;
__device__ __forceinline__ int s3m_ts(int J) { return J * S3M_NH + S3M_PAD; }
// flat item = w * J + j  ->  word index of its 8 heads.  rJ = 1.f / J: w = item / J as a float product (exact for item < 2^16, J <= 64;
// an integer division by the run-time J costs ~35 instructions, and the item loops of the backward need one per item)
__device__ __forceinline__ int s3m_item(int item, float rJ) { return item * S3M_NH + (int)(((float)item + 0.5f) * rJ) * S3M_PAD; }

// The head-mix loops (read the 8 heads of a table item, 8 x 8 FMAs, write them back in place) and a hazard found late in round 4
// (tools/determinism_stress.py): compiled freely, the 64 FMAs become 32 v_pk_fma_f32 on register pairs filled by two ds_read_b128 behind a
// PARTIAL s_waitcnt, with the first ds_write_b128 issued in the middle of them.  With two workgroups of the two-row forward tile on a CU
// (the LDS queue backed up) a few P' values per launch then came out wrong -- always the LOW half of a packed pair (an even head), a run of
// lanes of one wave, different ones every run; the tables before the mix were bit-stable, the ones after it were not, and one workgroup per
// CU never showed it.  Completing the stores before going on did NOT cure it; keeping every head's sum in its own scalar FMA chain (the
// empty asm below pins `s` per head, so no packed FMAs are formed) did: bit-identical over 12 runs at b = 16, all dilations, both operand
// forms.  The mechanism inside the packed-FMA sequence is not understood; the one-row kernel and the backward's mix never failed the same
// stress test but share the source pattern, so all three loops are written this way.  lds_store8_done: 8 consecutive floats -> LDS as two
// 16-byte stores, completed before the caller goes on, nothing scheduled across.
// S3_MIX_PIN (compile-time): 1 pins every head's sum of the three mix loops in its own scalar FMA chain and completes the stores before going on
// -- the round-4 cure, found empirically.  Round 5 separated the cures on the GPU (profiles/r05a_mix_variants.txt, r05e_stress_nofix.txt): with
// packed fp32 ops ON and the loops written freely the stress fails (431 k differing elements in 8 runs at b = 16); with packed ops OFF and the
// SAME free loops it is clean (b = 16 x 8 runs and b = 128 x 10 runs, every dilation, both operand forms).  The library ships without packed
// fp32 ops (build.py DEFAULT_FLAGS, gated by tools/isa_lint.py), so the pin is no longer needed and costs 0.5 % of the step: default 0.
// Build variant 'pin' (-DS3_MIX_PIN=1) keeps it for A/B runs.
#ifndef S3_MIX_PIN
#define S3_MIX_PIN 0
#endif
#if S3_MIX_PIN
#define S3_MIX_PIN_ASM(s) asm volatile("" : "+v"(s))
#else
#define S3_MIX_PIN_ASM(s) ((void)0)
#endif
__device__ __forceinline__ void lds_store8_done(float* p, const float (&v)[8]) {
#if S3_MIX_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int g = 0; g < 8; ++g) p[g] = v[g];
#if S3_MIX_PIN
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// fills pslot / ptok (thread 0) and returns after a barrier; `cnt` = &pslot[S3M_PLANES]
__device__ __forceinline__ void rowm_planes(const S3Args& a, int f, int y, int* pslot, int* ptok) {
    // closed form, one thread per plane (was a serial loop of thread 0): taps ta >= ta0 / tb >= tb0 reach a frame / row inside the grid
    const int ta0 = max(0, a.kf - 1 - f / a.df), tb0 = max(0, a.kh - 1 - y / a.dh);
    const int nb = a.kh - tb0, n = (a.kf - ta0) * nb;
    const int e = threadIdx.x;
    if (e < n) {
        const int ta = ta0 + e / nb, tb = tb0 + e % nb;
        pslot[e] = 1 + (ta * a.kh + tb) * a.kw;
        ptok[e] = 1 + ((f - (a.kf - 1 - ta) * a.df) * a.H + (y - (a.kh - 1 - tb) * a.dh)) * S3M_W;
    }
    if (e == 0) pslot[S3M_PLANES] = n;
    __syncthreads();
}
__device__ __forceinline__ RowM rowm_init(const S3Args& a, int b, int ry, const int* pslot, const int* ptok) {
    RowM r;
    r.nplanes = pslot[S3M_PLANES];
    r.J = a.kf * a.kh * a.kw + 1;
    r.TS = s3m_ts(r.J);
    r.lane = threadIdx.x & 63; r.wave = threadIdx.x >> 6; r.c = r.lane & 15; r.g4 = r.lane >> 4;
    r.tok0 = (size_t)b * a.ntok;
    r.iq = 1 + ry * S3M_W + r.c;
    r.qok = r.iq < a.ntok;
    r.pslot = pslot; r.ptok = ptok;
    r.vps = r.lane < r.nplanes ? pslot[r.lane] : 0;
    r.vpt = r.lane < r.nplanes ? ptok[r.lane] : 0;
    // Which tap (if any) links query c to each of the lane's 4 keys 4*g4 + r: the same for every plane, so all of the band
    // logic of the sweeps is decided here once.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = r.c - (4 * r.g4 + q);
        r.tsel[q] = -1;
#pragma unroll
        for (int tc = 0; tc < S3M_KW; ++tc)
            if (tc < a.kw && d == (a.kw - 1 - tc) * a.dw) r.tsel[q] = tc;
    }
    return r;
}

// Band scores of head h:  TAB[(c*J + slot)*NH + h] = mul * (frag row of query c) . (row of the slot's key) (+ bias[slot][h])
// for the <bos> slot and every tap of every valid plane.  S^T = ROWS . FRAG^T: the 16 key rows of a plane go straight from
// global memory into the A operand; the lane keeps the entries on the tap diagonals.  No LDS staging, no workgroup barrier.
__device__ __forceinline__ void mfma_band_scores(const S3Args& a, const RowM& r, const bf16_t* rows, int ldr, const bf16_t* frag,
                                                 int ldf, int h, float* TAB, float mul, const float* bias) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const bf16_t* qrow = frag + (r.tok0 + r.iq) * ldf + h * DH + r.g4 * 8;
    const bf16x8 qf0 = ldg8(qrow, r.qok), qf1 = ldg8(qrow + 32, r.qok);
    const bf16_t* kbase = rows + r.tok0 * ldr + h * DH + r.g4 * 8;                // + token * ld
    const int spb = r.c * r.TS + h;                                              // index of (query c, slot 0, head h)
    int sidx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sidx[q] = spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH;
    // sequence 0 is <bos> (token 0 for every MFMA row), then the valid planes; PF planes of key fragments are in flight
    constexpr int PF = S3M_PF;
    bf16x8 kq0[PF], kq1[PF];
    auto issue = [&](int sq, bf16x8& d0, bf16x8& d1) {
        if (sq > r.nplanes) { d0 = d1 = bf16x8{}; return; }
        const int tok = sq == 0 ? 0 : r.ptok[sq - 1] + r.c;
        const bf16_t* kp = kbase + (size_t)tok * ldr;
        const bool ok = tok < a.ntok;
        d0 = ldg8(kp, ok); d1 = ldg8(kp + 32, ok);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) issue(i, kq0[i], kq1[i]);
    for (int sq = 0; sq <= r.nplanes; ++sq) {
        const bf16x8 k0 = kq0[0], k1 = kq1[0];
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i) { kq0[i] = kq0[i + 1]; kq1[i] = kq1[i + 1]; }
        issue(sq + PF, kq0[PF - 1], kq1[PF - 1]);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf0, sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf1, sc, 0, 0, 0);
        if (sq == 0) {
            if (r.g4 == 0 && r.qok) TAB[spb] = sc[0] * mul + (bias ? bias[h] : 0.f);
        } else if (r.qok) {
            const int jb = r.pslot[sq - 1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r.tsel[q] >= 0) TAB[sidx[q] + jb * NH] = sc[q] * mul + (bias ? bias[(jb + r.tsel[q]) * NH + h] : 0.f);
        }
    }
}

// Band apply of head g:  O[query c][d] = TAB[c][0][g] * rows[<bos>][d] + sum over planes / taps TAB[c][slot][g] * rows[key][d]
// as O^T = ROWS^T . TAB^T: the rows of TWO planes (32 keys) are staged in the wave-private 4 KiB LDS tile and read back
// transposed (ds_read_b64_tr_b16) as the A operand; the banded coefficients of the lane's 8 key slots come from TAB.
// Lane (c, g4) ends with O[db][q] = out[query c][db*16 + 4*g4 + q].
// The same scores with the key rows STAGED through a wave-private LDS tile ([16 keys][64] bf16, 2 KiB): every lane fetches two
// 16-byte pieces of FULL 128-byte row slices (8 lanes per row) instead of fragment-shaped 64-byte halves (4 lanes per row, two
// instructions per row), then reads its two MFMA fragments out of LDS.  Phase ablation of the forward (tuning key 9, dilation 1,
// b = 64): scores 242 us, apply (which always loaded full lines) 166 us for the same number of bytes.
template <bool F16 = false>     // F16: rows / frag hold fp16 values and the products run on the fp16 MFMA (forward core of 'bf16x3-fwd')
__device__ __forceinline__ void mfma_band_scores_staged(const S3Args& a, const RowM& r, const bf16_t* rows, int ldr, const bf16_t* frag,
                                                        int ldf, int h, float* TAB, float mul, const float* bias, char* tile) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const bf16_t* qrow = frag + (r.tok0 + r.iq) * ldf + h * DH + r.g4 * 8;
    const bf16x8 qf0 = ldg8(qrow, r.qok), qf1 = ldg8(qrow + 32, r.qok);
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const bf16_t* kbase = rows + r.tok0 * ldr + h * DH + gc * 8;                  // + token * ld
    const int spb = r.c * r.TS + h;
    int sidx[4], wbase[4], wmul[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sidx[q] = spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH;
        const bool on = r.tsel[q] >= 0 && r.qok;
        wbase[q] = on ? sidx[q] : r.c * r.TS + r.J * NH + r.g4;          // (the 4 pad words of query c's table row: one per lane group)
        wmul[q] = on ? NH : 0;
    }
    const int w0 = vt_off(r8, gc), w1 = vt_off(r8 + 8, gc);
    const int f0 = vt_off(r.c, r.g4), f1 = vt_off(r.c, 4 + r.g4);
    constexpr int PF = S3M_PF;
    static_assert(PF == 3, "the sweep below names its three register sets");
    auto issue = [&](int sq, uint4& d0, uint4& d1) {
#if S3M_BRANCHFREE
        // (no load under a branch: rows that do not exist read token 0 and are zeroed by a select)
        const int base = (sq == 0 || sq > r.nplanes) ? -1 : r.ptok[sq - 1];        // sequence 0: every row is token 0 (<bos>)
        const int t0 = base < 0 ? 0 : base + r8, t1 = base < 0 ? 0 : base + r8 + 8;
        const bool live = sq <= r.nplanes;
        const uint4 v0 = *reinterpret_cast<const uint4*>(kbase + (size_t)(t0 < a.ntok ? t0 : 0) * ldr);
        const uint4 v1 = *reinterpret_cast<const uint4*>(kbase + (size_t)(t1 < a.ntok ? t1 : 0) * ldr);
        d0 = (live && t0 < a.ntok) ? v0 : make_uint4(0, 0, 0, 0);
        d1 = (live && t1 < a.ntok) ? v1 : make_uint4(0, 0, 0, 0);
#else
        if (sq > r.nplanes) { d0 = d1 = make_uint4(0, 0, 0, 0); return; }
        const int base = sq == 0 ? -1 : r.ptok[sq - 1];                          // sequence 0: every row is token 0 (<bos>)
        const int t0 = base < 0 ? 0 : base + r8, t1 = base < 0 ? 0 : base + r8 + 8;
        d0 = t0 < a.ntok ? *reinterpret_cast<const uint4*>(kbase + (size_t)t0 * ldr) : make_uint4(0, 0, 0, 0);
        d1 = t1 < a.ntok ? *reinterpret_cast<const uint4*>(kbase + (size_t)t1 * ldr) : make_uint4(0, 0, 0, 0);
#endif
    };
    // three planes of rows in flight in three NAMED register sets, refilled in place as soon as their rows sit in the tile: the loop is
    // unrolled by three so no set ever has to be moved (the rotating form spent 16 v_mov per plane, a sixth of the sweep's instructions)
    uint4 sa0, sa1, sb0, sb1, sc0, sc1;
    issue(0, sa0, sa1); issue(1, sb0, sb1); issue(2, sc0, sc1);
    auto step = [&](int sq, uint4& d0, uint4& d1) {
        *reinterpret_cast<uint4*>(tile + w0) = d0;
        *reinterpret_cast<uint4*>(tile + w1) = d1;
        issue(sq + PF, d0, d1);
        __builtin_amdgcn_wave_barrier();                                          // LDS is in-order per wave: the tile is complete
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(tile + f0), k1 = *reinterpret_cast<const bf16x8*>(tile + f1);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = mfma16<F16>(k0, qf0, sc);
        sc = mfma16<F16>(k1, qf1, sc);
        __builtin_amdgcn_wave_barrier();
        if (sq == 0) {
            if (r.g4 == 0 && r.qok) TAB[spb] = sc[0] * mul + (bias ? bias[h] : 0.f);
        } else if (!bias) {
            // band scatter without exec masking: an accumulator entry that lies on no tap diagonal goes to the lane's own pad word of the
            // table (wbase = the pad, wmul = 0), the others to (slot jb + tap, head h): 4 x (mad, mul, ds_write) instead of 4 masked blocks
            const int jb = r.pslot[sq - 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) TAB[wbase[q] + jb * wmul[q]] = sc[q] * mul;
        } else if (r.qok) {
            const int jb = r.pslot[sq - 1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r.tsel[q] >= 0) TAB[sidx[q] + jb * NH] = sc[q] * mul + bias[(jb + r.tsel[q]) * NH + h];
        }
    };
    for (int sq = 0; sq <= r.nplanes; sq += 3) {
        step(sq, sa0, sa1);
        if (sq + 1 <= r.nplanes) step(sq + 1, sb0, sb1);
        if (sq + 2 <= r.nplanes) step(sq + 2, sc0, sc1);
    }
}

template <bool F16 = false>
__device__ __forceinline__ void mfma_band_apply(const S3Args& a, const RowM& r, const bf16_t* rows, int ldr, int g, const float* TAB,
                                                char* tile, f32x4 (&O)[4]) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const int spb = r.c * r.TS + g;
    {   // <bos> slot
        const float p0 = r.qok ? TAB[spb] : 0.f;
        const bf16_t* vb = rows + r.tok0 * ldr + g * DH + 4 * r.g4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const uint2 u = *reinterpret_cast<const uint2*>(vb + db * 16);
            O[db] = f32x4{p0 * lo_t<F16>(u.x), p0 * hi_t<F16>(u.x), p0 * lo_t<F16>(u.y), p0 * hi_t<F16>(u.y)};
        }
    }
    // staging map of a chunk (two planes = 32 rows x 8 sixteen-byte pieces, 4 per lane): piece i of this lane is row
    // (lane >> 3) + 8 i, i.e. pieces 0, 1 belong to the first plane and 2, 3 to the second -- all offsets are fixed
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const bf16_t* vbase = rows + r.tok0 * ldr + g * DH + gc * 8;
    int woff[4], sidx[4], troff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = vt_off(r8 + 8 * i, gc);
#pragma unroll
    for (int q = 0; q < 4; ++q) sidx[q] = spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH;
    {   // transposing-read offsets: rows r0 and 16 + r0 (same swizzle since 16 & 7 == 0: + 2048 bytes), column block db
        const int r0 = 4 * r.g4 + (r.c >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 16 + ((r.c & 3) << 2);
            troff[db] = vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8;
        }
    }
    uint4 st[4];
    auto fetch = [&](int pi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pj = pi + (i >> 1);
            const int tok = pj < r.nplanes ? r.ptok[pj] + r8 + 8 * (i & 1) : a.ntok;
#if S3M_BRANCHFREE
            const uint4 v = *reinterpret_cast<const uint4*>(vbase + (size_t)(tok < a.ntok ? tok : 0) * ldr);
            st[i] = tok < a.ntok ? v : make_uint4(0, 0, 0, 0);
#else
            st[i] = tok < a.ntok ? *reinterpret_cast<const uint4*>(vbase + (size_t)tok * ldr) : make_uint4(0, 0, 0, 0);
#endif
        }
    };
    fetch(0);
    for (int pi = 0; pi < r.nplanes; pi += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + woff[i]) = st[i];
        const int jb0 = r.pslot[pi] * NH, jb1 = pi + 1 < r.nplanes ? r.pslot[pi + 1] * NH : -1;
        if (pi + 2 < r.nplanes) fetch(pi + 2);                                    // the next chunk's rows are in flight below
        float pf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = r.tsel[j] >= 0 && r.qok;
            pf[j] = on ? TAB[sidx[j] + jb0] : 0.f;
            pf[4 + j] = (on && jb1 >= 0) ? TAB[sidx[j] + jb1] : 0.f;
        }
        const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(pack2_t<F16>(pf[0], pf[1]), pack2_t<F16>(pf[2], pf[3]),
                                                                 pack2_t<F16>(pf[4], pf[5]), pack2_t<F16>(pf[6], pf[7])));
        __builtin_amdgcn_wave_barrier();                                          // LDS is in-order per wave: the tile is complete
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db] + 2048));
            const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            O[db] = mfma16<F16>(__builtin_bit_cast(bf16x8, v8), pb, O[db]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- round 6: the two single-row sweeps with their address arithmetic taken out of the plane loop ----------------------------------------------
// The ISA of mfma_band_scores_staged / mfma_band_apply spent, per plane and wave: an LDS read of the plane's slot / token (a wave-uniform value
// fetched through a VGPR), four v_mul_lo_u32 + shifts + adds for the four table addresses, two v_mad_i64_i32 + v_lshl_add_u64 for the two row
// addresses and an exec-masked branch around each load -- about half of the sweep's ~80 instructions, on a chip where a VALU instruction costs
// ~4 cycles of its SIMD (profiles/r06w_geglu_pmc.txt).  Here: the plane lists sit in one register each (plane p in lane p, v_readlane with the
// loop counter), the key rows are addressed as a wave-uniform 64-bit base + a 32-bit byte offset = (plane's first token) x (row bytes) [scalar]
// + a per-lane constant, a plane whose 16 rows all exist (every plane but the last row of a sample) loads without a mask, and a table address
// is ONE v_mad_u32_u24 (slot x 32 bytes or x 0, + the lane's base address).  Same loads, same MFMAs, same stores, same values.
typedef __attribute__((address_space(3))) float lds_f32_t;
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void lds_st32(unsigned addr, float v) { *reinterpret_cast<lds_f32_t*>((size_t)addr) = v; }
__device__ __forceinline__ float lds_ld32(unsigned addr) { return *reinterpret_cast<const lds_f32_t*>((size_t)addr); }
__device__ __forceinline__ uint4 ldg_u4(const char* sb, unsigned off) { return *reinterpret_cast<const uint4*>(sb + off); }

template <bool F16>
__device__ __forceinline__ void mfma_band_scores_fast(const S3Args& a, const RowM& r, const bf16_t* rows, int ldr, const bf16_t* frag, int ldf,
                                                      int h, float* TAB, float mul, char* tile) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const bf16_t* qrow = frag + (r.tok0 + r.iq) * ldf + h * DH + r.g4 * 8;
    const bf16x8 qf0 = ldg8(qrow, r.qok), qf1 = ldg8(qrow + 32, r.qok);
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const char* sb = reinterpret_cast<const char*>(rows + r.tok0 * ldr + __builtin_amdgcn_readfirstlane(h) * DH);   // wave-uniform (h = the wave's index: said so, it stays in scalar registers)
    const unsigned ldb = (unsigned)ldr * 2u;
    const unsigned vb = (unsigned)gc * 16u, v0 = (unsigned)r8 * ldb + vb, v1 = v0 + 8u * ldb;
    const unsigned tab0 = lds_byte_addr(TAB);
    const int spb = r.c * r.TS + h;
    unsigned wb4[4], wm4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool on = r.tsel[q] >= 0 && r.qok;
        wb4[q] = tab0 + 4u * (unsigned)(on ? spb + r.tsel[q] * NH : r.c * r.TS + r.J * NH + r.g4);   // (off: the lane's own pad word of the table row)
        wm4[q] = on ? 4u * NH : 0u;
    }
    const int w0 = vt_off(r8, gc), w1 = vt_off(r8 + 8, gc);
    const int f0 = vt_off(r.c, r.g4), f1 = vt_off(r.c, 4 + r.g4);
    constexpr int PF = S3M_PF;
    static_assert(PF == 3, "the sweep below names its three register sets");
    auto issue = [&](int sq, uint4& d0, uint4& d1) {
        if (sq > r.nplanes) return;                                                          // (past the end: the register set is never consumed)
        if (sq == 0) { d0 = ldg_u4(sb, vb); d1 = d0; return; }                               // sequence 0: every row is token 0 (<bos>)
        const int base = __builtin_amdgcn_readlane(r.vpt, sq - 1);
        const unsigned so = (unsigned)base * ldb;
        if (base + 16 <= a.ntok) { d0 = ldg_u4(sb, so + v0); d1 = ldg_u4(sb, so + v1); }
        else {
            d0 = base + r8 < a.ntok ? ldg_u4(sb, so + v0) : make_uint4(0, 0, 0, 0);
            d1 = base + r8 + 8 < a.ntok ? ldg_u4(sb, so + v1) : make_uint4(0, 0, 0, 0);
        }
    };
    uint4 sa0 = make_uint4(0, 0, 0, 0), sa1 = sa0, sb0 = sa0, sb1 = sa0, sc0 = sa0, sc1 = sa0;
    issue(0, sa0, sa1); issue(1, sb0, sb1); issue(2, sc0, sc1);
    auto step = [&](int sq, uint4& d0, uint4& d1) {
        *reinterpret_cast<uint4*>(tile + w0) = d0;
        *reinterpret_cast<uint4*>(tile + w1) = d1;
        issue(sq + PF, d0, d1);
        __builtin_amdgcn_wave_barrier();
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(tile + f0), k1 = *reinterpret_cast<const bf16x8*>(tile + f1);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = mfma16<F16>(k0, qf0, sc);
        sc = mfma16<F16>(k1, qf1, sc);
        __builtin_amdgcn_wave_barrier();
        if (sq == 0) {
            if (r.g4 == 0 && r.qok) lds_st32(tab0 + 4u * (unsigned)spb, sc[0] * mul);
        } else {
            const unsigned jb = (unsigned)__builtin_amdgcn_readlane(r.vps, sq - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) lds_st32(__umul24(jb, wm4[q]) + wb4[q], sc[q] * mul);
        }
    };
    for (int sq = 0; sq <= r.nplanes; sq += 3) {
        step(sq, sa0, sa1);
        if (sq + 1 <= r.nplanes) step(sq + 1, sb0, sb1);
        if (sq + 2 <= r.nplanes) step(sq + 2, sc0, sc1);
    }
}

template <bool F16>
__device__ __forceinline__ void mfma_band_apply_fast(const S3Args& a, const RowM& r, const bf16_t* rows, int ldr, int g, const float* TAB,
                                                     char* tile, f32x4 (&O)[4]) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const int spb = r.c * r.TS + g;
    {   // <bos> slot
        const float p0 = r.qok ? TAB[spb] : 0.f;
        const bf16_t* vb = rows + r.tok0 * ldr + g * DH + 4 * r.g4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const uint2 u = *reinterpret_cast<const uint2*>(vb + db * 16);
            O[db] = f32x4{p0 * lo_t<F16>(u.x), p0 * hi_t<F16>(u.x), p0 * lo_t<F16>(u.y), p0 * hi_t<F16>(u.y)};
        }
    }
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const char* sb = reinterpret_cast<const char*>(rows + r.tok0 * ldr + __builtin_amdgcn_readfirstlane(g) * DH);   // wave-uniform
    const unsigned ldb = (unsigned)ldr * 2u;
    const unsigned v0 = (unsigned)r8 * ldb + (unsigned)gc * 16u, v1 = v0 + 8u * ldb;
    const unsigned tab0 = lds_byte_addr(TAB);
    int woff[4], troff[4];
    unsigned sidx4[4];
    bool on[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = vt_off(r8 + 8 * i, gc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        on[q] = r.tsel[q] >= 0 && r.qok;
        sidx4[q] = tab0 + 4u * (unsigned)(spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH);
    }
    {
        const int r0 = 4 * r.g4 + (r.c >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 16 + ((r.c & 3) << 2);
            troff[db] = vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8;
        }
    }
    uint4 st[4];
    auto fetch2 = [&](int pj, uint4& d0, uint4& d1) {                                       // the 16 rows of plane pj: this lane's two pieces
        if (pj >= r.nplanes) { d0 = d1 = make_uint4(0, 0, 0, 0); return; }                   // (the odd plane of the last pair: its rows must be zeros)
        const int base = __builtin_amdgcn_readlane(r.vpt, pj);
        const unsigned so = (unsigned)base * ldb;
        if (base + 16 <= a.ntok) { d0 = ldg_u4(sb, so + v0); d1 = ldg_u4(sb, so + v1); }
        else {
            d0 = base + r8 < a.ntok ? ldg_u4(sb, so + v0) : make_uint4(0, 0, 0, 0);
            d1 = base + r8 + 8 < a.ntok ? ldg_u4(sb, so + v1) : make_uint4(0, 0, 0, 0);
        }
    };
    fetch2(0, st[0], st[1]); fetch2(1, st[2], st[3]);
    for (int pi = 0; pi < r.nplanes; pi += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + woff[i]) = st[i];
        const unsigned jb0 = 4u * NH * (unsigned)__builtin_amdgcn_readlane(r.vps, pi);
        const bool two = pi + 1 < r.nplanes;
        const unsigned jb1 = two ? 4u * NH * (unsigned)__builtin_amdgcn_readlane(r.vps, pi + 1) : jb0;
        if (pi + 2 < r.nplanes) { fetch2(pi + 2, st[0], st[1]); fetch2(pi + 3, st[2], st[3]); }   // the next chunk's rows are in flight below
        float pf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = lds_ld32(sidx4[j] + jb0), x1 = lds_ld32(sidx4[j] + jb1);
            pf[j] = on[j] ? x0 : 0.f;
            pf[4 + j] = (on[j] && two) ? x1 : 0.f;
        }
        const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(pack2_t<F16>(pf[0], pf[1]), pack2_t<F16>(pf[2], pf[3]),
                                                                 pack2_t<F16>(pf[4], pf[5]), pack2_t<F16>(pf[6], pf[7])));
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db] + 2048));
            const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            O[db] = mfma16<F16>(__builtin_bit_cast(bf16x8, v8), pb, O[db]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// fp32 softmax over the J slots of every (w, h) of TAB, in place (4 lanes split j); masked slots hold NEG_MAX
// gst (optional): [W][NH][4] floats of this query row in the statistics array: (row max, 1 / row sum) of queries w < wvalid go there
// pre: the table holds raw q . k products and the softmax scale is applied here (one multiply per slot instead of one per product
// in the score sweep; the same fp32 multiply of the same value, so nothing changes bit-wise).  Kernels with a bias table scale in the sweep.
__device__ __forceinline__ void rowm_softmax(float* TAB, int J, float* gst = nullptr, int wvalid = 0, float pre = 1.f) {
    constexpr int NH = S3M_NH;
    const int t = threadIdx.x, cc = t & 3, wh = t >> 2, h = wh % NH, w = wh / NH;
    TAB += w * s3m_ts(J) + h;                                                    // (query w, slot 0, head h); slot j at + j * NH
    if (J <= 48) {
        // the thread's <= 12 slots in registers: one batch of LDS reads and one of writes instead of three dependent
        // read (-modify-write) passes of 12 round trips each; same operations in the same order -> bit-identical
        float v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { const int j = cc + 4 * k; v[k] = j < J ? TAB[j * NH] * pre : NEG_MAX; }
        float m = NEG_MAX;
#pragma unroll
        for (int k = 0; k < 12; ++k) m = fmaxf(m, v[k]);
        m = quad_max(m);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float e = cc + 4 * k < J ? __expf(v[k] - m) : 0.f;
            v[k] = e;
            if (cc + 4 * k < J) sum += e;
        }
        sum = quad_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int k = 0; k < 12; ++k) { const int j = cc + 4 * k; if (j < J) TAB[j * NH] = v[k] * inv; }
        if (gst && cc == 0 && w < wvalid) *reinterpret_cast<float2*>(gst + (w * NH + h) * 4) = make_float2(m, inv);
        return;
    }
    float m = NEG_MAX;
    for (int j = cc; j < J; j += 4) m = fmaxf(m, TAB[j * NH] * pre);
    m = quad_max(m);
    float sum = 0.f;
    for (int j = cc; j < J; j += 4) {
        const int idx = j * NH;
        const float e = __expf(TAB[idx] * pre - m);
        TAB[idx] = e;
        sum += e;
    }
    sum = quad_sum(sum);
    const float inv = 1.f / sum;
    for (int j = cc; j < J; j += 4) TAB[j * NH] *= inv;
    if (gst && cc == 0 && w < wvalid) *reinterpret_cast<float2*>(gst + (w * NH + h) * 4) = make_float2(m, inv);
}

// F16: a.q / a.k / a.v hold fp16 values, the score and apply products run on the fp16 MFMA, and o leaves as a bf16 hi + lo pair
// (a.ol): the forward Sparse3DNA core of the 'bf16x3-fwd' precision mode.
template <bool F16>
__global__ __launch_bounds__(512, 2) void s3_fwd_mfma_kernel(S3Args a) {
    constexpr int NH = S3M_NH, DH = S3M_DH, W = S3M_W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int J = a.kf * a.kh * a.kw + 1;
    float* SP = reinterpret_cast<float*>(smem);                                  // [W][J * NH + 4]  (s3m_ts)
    char* vt_base = smem + (size_t)W * s3m_ts(J) * sizeof(float);                // 8 wave-private [32][64] bf16 tiles
    __shared__ float wsh[64];
    __shared__ int pslot[S3M_PLANES + 1], ptok[S3M_PLANES + 1];
    const int t = threadIdx.x;
    const int rows = a.F * a.H;
    const int bid = xcd_row_id();
    const int b = bid / rows, r2 = bid % rows;
    int f, y;
    s3m_row_order(a, r2, f, y);
    const int ry = f * a.H + y;
    if (t < NH * NH) wsh[t] = a.wth[t];
    if (ry == 0) {                                                               // <bos> output row = its own value
        for (int e = t; e < NH * DH; e += blockDim.x) {
            const bf16_t raw = a.v[((size_t)b * a.ntok) * a.ld + e];
            if (F16) {
                bf16_t hi, lo;
                f2bf_hilo((float)__builtin_bit_cast(_Float16, raw), hi, lo);
                if (a.o) a.o[((size_t)b * a.ntok) * a.ldo + e] = hi;     // (o == NULL: the fp16 copy alone -- the fp16-gradient backward reads nothing else)
                if (a.ol) a.ol[((size_t)b * a.ntok) * a.ldo + e] = a.ol_f16 ? raw : lo;
            } else a.o[((size_t)b * a.ntok) * a.ldo + e] = raw;
        }
    }
    if (ry * W + 1 >= a.ntok) return;                                            // whole row beyond the sequence (uniform)
    for (int e = t; e < W * s3m_ts(J); e += blockDim.x) SP[e] = NEG_MAX;
    rowm_planes(a, f, y, pslot, ptok);
    const RowM r = rowm_init(a, b, ry, pslot, ptok);
    if (!(a.dbg & 1)) {
        if (!F16 && (a.dbg & 8)) mfma_band_scores(a, r, a.k, a.ld, a.q, a.ld, r.wave, SP, a.scale, a.bias);
        else mfma_band_scores_staged<F16>(a, r, a.k, a.ld, a.q, a.ld, r.wave, SP, a.scale, a.bias, vt_base + r.wave * 4096);
    }
    __syncthreads();
    if (!(a.dbg & 2)) rowm_softmax(SP, J);
    __syncthreads();
    // talking heads: P'[g] = sum_h Wth[g][h] P[h] per (w, j), in place
    float wr[64];                              // the 8 x 8 mix matrix in registers for the loop below: read from LDS inside it, every output row
#pragma unroll                                     // waited for its own LDS round trip (the stores to the table may alias wsh for the compiler)
    for (int k = 0; k < 64; ++k) wr[k] = wsh[k];
    for (int item = t; item < W * J && !(a.dbg & 2); item += blockDim.x) {
        float pv[8], out[8];
        const int ib = s3m_item(item, 1.f / (float)J);
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) pv[hh] = SP[ib + hh];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float s = 0.f;
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) s += wr[g * NH + hh] * pv[hh];
            S3_MIX_PIN_ASM(s);                   // (one head at a time: see lds_store8_done)
            out[g] = s;
        }
        lds_store8_done(SP + ib, out);
    }
    __syncthreads();
    if (!(a.dbg & 4)) {
        const int g = r.wave;
        f32x4 O[4];
        mfma_band_apply<F16>(a, r, a.v, a.ld, g, SP, vt_base + r.wave * 4096, O);
        if (r.qok) {
            const size_t go = (r.tok0 + r.iq) * a.ldo + g * DH + 4 * r.g4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const uint32_t h01 = pack2_rne(O[db][0], O[db][1]), h23 = pack2_rne(O[db][2], O[db][3]);
                if (!F16 || a.o) *reinterpret_cast<uint2*>(a.o + go + db * 16) = make_uint2(h01, h23);
                if (F16 && a.ol)
                    *reinterpret_cast<uint2*>(a.ol + go + db * 16) = a.ol_f16 ?
                        make_uint2(pack2_f16_sat(O[db][0], O[db][1]), pack2_f16_sat(O[db][2], O[db][3])) :
                        make_uint2(pack2_rne(O[db][0] - lo_f(h01), O[db][1] - hi_f(h01)), pack2_rne(O[db][2] - lo_f(h23), O[db][3] - hi_f(h23)));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-row tiles of the MFMA kernels (tuning key 16).  One workgroup = ROWS query rows (f, y0 + i dh), i = 0 .. ROWS-1, of one
// residue class of y modulo the dilation: their key rows are (f - a df, y0 + (i - b) dh), i.e. per tap frame only ROWS + kh - 1
// distinct rows instead of ROWS * kh.  Every key / value row is fetched from L2 and staged in the wave's LDS tile ONCE per tile and
// its MFMA fragments are read ONCE; the (up to kh) query rows that tap it reuse them from registers: per query row the L2 -> CU
// gathers, the ds_write_b128 staging stores (13 LDS cycles each) and the fragment reads drop by ROWS kh / (ROWS + kh - 1) = 2x at
// ROWS = 4, kh = 3 (1.5x at ROWS = 2).  The per-row score tables, the softmax and the head mix are the single-row kernel's, looped
// over the rows: the score entries are bit-identical, the apply pass pairs its key rows differently inside an MFMA (fp32 summation
// order only).  LDS: ROWS x 23.8 KiB of fp32 tables + 8 x 4 KiB tiles: two workgroups per CU at ROWS = 2, one at ROWS = 4.
// ---------------------------------------------------------------------------------------------
constexpr int S3T_MAXK = 128;                    // key rows of a tile: kf * (kh + ROWS - 1)
constexpr int S3T_AUTO_ROWS = 2;                 // tuning key 16 = 0.  A/B at b = 128 (profiles/r04a_attn.txt): 2 rows -14..-16 % on every dilation
                                                 // (two workgroups per CU as before); 4 rows +4..+22 % SLOWER (95 KiB of tables: one workgroup per CU)
template <int ROWS>
struct TileM {
    int nk, nrows, J, TS, WTS, lane, wave, c, g4;
    size_t tok0;
    int iq[ROWS];                                // token row of query c in tile row i
    bool qok[ROWS];
    int tsel[4];
    const int *ktok, *kmeta;                     // key-row list: token row of key 0; meta = tap frame ta << 8 | (m + 64), y_key = y0 + m dh
    int vkt, vkm;                                // the same list with entry e in LANE e of a register (nk <= 64: tile_band_scores_fast)
};
// tile t2 of a sample -> (frame, first row y0); the order keeps a residue class of y together (see s3m_row_order)
template <int ROWS>
__device__ __forceinline__ void s3t_tile_order(const S3Args& a, int t2, int& f, int& y0) {
    const int ng = a.H / (a.dh * ROWS);          // tiles per residue class and frame
    int cl, gi;
    if (a.ymajor) { const int per = ng * a.F; cl = t2 / per; const int rem = t2 % per; gi = rem / a.F; f = rem % a.F; }
    else { const int per = a.H / ROWS; f = t2 / per; const int rem = t2 % per; cl = rem / ng; gi = rem % ng; }
    y0 = cl + gi * ROWS * a.dh;
}
// The key-row list in closed form (one thread per entry instead of a serial loop of thread 0 with 511 threads waiting at the
// barrier): frame taps ta >= ta0 have fr = f - (kf - 1 - ta) df >= 0, rows m >= m0 have y0 + m dh >= 0, so the valid entries are the
// rectangle [ta0, kf) x [m0, ROWS) in (ta, m) order -- the order the serial loop produced.
template <int ROWS>
__device__ __forceinline__ void s3t_keylist(const S3Args& a, int f, int y0, int* ktok, int* kmeta, int* kcnt) {
    const int ta0 = max(0, a.kf - 1 - f / a.df);                  // (kf - 1 - ta) df <= f
    const int m0 = max(-(a.kh - 1), -(y0 / a.dh));                // y0 + m dh >= 0
    const int nm = ROWS - m0, n = (a.kf - ta0) * nm;
    const int e = threadIdx.x;
    if (e < n) {
        const int ta = ta0 + e / nm, m = m0 + e % nm;
        ktok[e] = 1 + ((f - (a.kf - 1 - ta) * a.df) * a.H + y0 + m * a.dh) * S3M_W;
        kmeta[e] = (ta << 8) | (m + 64);
    }
    if (e == 0) *kcnt = n;
    __syncthreads();
}
template <int ROWS>
__device__ __forceinline__ TileM<ROWS> s3t_init(const S3Args& a, int b, int f, int y0, int nrows, const int* ktok, const int* kmeta, int nk) {
    TileM<ROWS> r;
    r.nk = nk; r.nrows = nrows;
    r.J = a.kf * a.kh * a.kw + 1; r.TS = s3m_ts(r.J); r.WTS = S3M_W * r.TS;
    r.lane = threadIdx.x & 63; r.wave = threadIdx.x >> 6; r.c = r.lane & 15; r.g4 = r.lane >> 4;
    r.tok0 = (size_t)b * a.ntok;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        r.iq[i] = 1 + (f * a.H + y0 + i * a.dh) * S3M_W + r.c;
        r.qok[i] = i < nrows && r.iq[i] < a.ntok;
    }
    r.ktok = ktok; r.kmeta = kmeta;
    r.vkt = r.lane < nk ? ktok[r.lane] : 0;
    r.vkm = r.lane < nk ? kmeta[r.lane] : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = r.c - (4 * r.g4 + q);
        r.tsel[q] = -1;
#pragma unroll
        for (int tc = 0; tc < S3M_KW; ++tc)
            if (tc < a.kw && d == (a.kw - 1 - tc) * a.dw) r.tsel[q] = tc;
    }
    return r;
}
// 16-byte load from a row that always exists, zero-filled by a select (a load under a branch costs the in-order wait counts)
#ifndef S3T_BRANCHFREE
#define S3T_BRANCHFREE 1
#endif
__device__ __forceinline__ uint4 ldg16_sel(const bf16_t* p, bool ok) {
#if S3T_BRANCHFREE
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    return ok ? v : make_uint4(0, 0, 0, 0);
#else
    return ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
#endif
}

// Band scores of head h for every row of the tile: TAB_i[(c, slot, h)] = mul * (frag row of query c of tile row i) . (key row) (+ bias),
// key rows staged once through the wave's [16][64] tile (see mfma_band_scores_staged) and multiplied with the fragments of the
// (up to kh) tile rows that tap them.
template <int ROWS, bool F16, int PF>
__device__ __forceinline__ void tile_band_scores(const S3Args& a, const TileM<ROWS>& r, const bf16_t* rows, int ldr, const bf16_t* frag,
                                                 int ldf, int h, float* TAB, float mul, const float* bias, char* tile) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    bf16x8 qf0[ROWS], qf1[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const bf16_t* qrow = frag + (r.tok0 + (r.qok[i] ? r.iq[i] : 0)) * ldf + h * DH + r.g4 * 8;
        qf0[i] = __builtin_bit_cast(bf16x8, ldg16_sel(qrow, r.qok[i]));
        qf1[i] = __builtin_bit_cast(bf16x8, ldg16_sel(qrow + 32, r.qok[i]));
    }
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const bf16_t* kbase = rows + r.tok0 * ldr + h * DH + gc * 8;                  // + token * ld
    const int spb = r.c * r.TS + h;
    const int wpad = r.c * r.TS + r.J * NH + r.g4;                               // the lane's pad word of query c's table row
    int sidx[4], wbase[4], wmul[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sidx[q] = spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH;
        wbase[q] = r.tsel[q] >= 0 ? sidx[q] : wpad;
        wmul[q] = r.tsel[q] >= 0 ? NH : 0;
    }
    const int w0 = vt_off(r8, gc), w1 = vt_off(r8 + 8, gc);
    const int f0 = vt_off(r.c, r.g4), f1 = vt_off(r.c, 4 + r.g4);
    auto issue = [&](int sq, uint4& d0, uint4& d1) {                             // sequence 0: every row is token 0 (<bos>)
        const int base = (sq == 0 || sq > r.nk) ? -1 : r.ktok[sq - 1];
        const int t0 = base < 0 ? 0 : base + r8, t1 = base < 0 ? 0 : base + r8 + 8;
        const bool live = sq <= r.nk;
        d0 = ldg16_sel(kbase + (size_t)(t0 < a.ntok ? t0 : 0) * ldr, live && t0 < a.ntok);
        d1 = ldg16_sel(kbase + (size_t)(t1 < a.ntok ? t1 : 0) * ldr, live && t1 < a.ntok);
    };
    // one key row: stage, read its two fragments once, multiply with the fragments of the (up to kh) tile rows that tap it; the
    // register set is refilled in place (see mfma_band_scores_staged: named sets, no rotation moves)
    auto step = [&](int sq, uint4& d0, uint4& d1) {
        *reinterpret_cast<uint4*>(tile + w0) = d0;
        *reinterpret_cast<uint4*>(tile + w1) = d1;
        issue(sq + PF, d0, d1);
        __builtin_amdgcn_wave_barrier();                                          // LDS is in-order per wave: the tile is complete
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(tile + f0), k1 = *reinterpret_cast<const bf16x8*>(tile + f1);
        // all MFMAs of the step first, then the scatters: the result latency of one row's products is covered by the next row's
        f32x4 sc[ROWS];
        if (sq == 0) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                if (i >= r.nrows) continue;
                sc[i] = mfma16<F16>(k1, qf1[i], mfma16<F16>(k0, qf0[i], f32x4{0.f, 0.f, 0.f, 0.f}));
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
                if (i < r.nrows && r.g4 == 0 && r.qok[i]) TAB[i * r.WTS + spb] = sc[i][0] * mul + (bias ? bias[h] : 0.f);
        } else {
            const int meta = r.kmeta[sq - 1], ta = meta >> 8, m = (meta & 255) - 64;
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int tb = m - i + a.kh - 1;
                if (i >= r.nrows || tb < 0 || tb >= a.kh) continue;              // (wave-uniform)
                sc[i] = mfma16<F16>(k1, qf1[i], mfma16<F16>(k0, qf0[i], f32x4{0.f, 0.f, 0.f, 0.f}));
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int tb = m - i + a.kh - 1;
                if (i >= r.nrows || tb < 0 || tb >= a.kh) continue;
                const int jb = 1 + (ta * a.kh + tb) * a.kw;
                if (!bias) {                                                     // (unmasked band scatter: see mfma_band_scores_staged)
                    const int wb = r.qok[i] ? 0 : 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        TAB[i * r.WTS + (wb ? wpad : wbase[q]) + jb * (wb ? 0 : wmul[q])] = sc[i][q] * mul;
                } else if (r.qok[i]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (r.tsel[q] >= 0)
                            TAB[i * r.WTS + sidx[q] + jb * NH] = sc[i][q] * mul + bias[(jb + r.tsel[q]) * NH + h];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    static_assert(PF == 3 || PF == 4, "named register sets");
    uint4 sa0, sa1, sb0, sb1, sc0, sc1, sd0, sd1;
    issue(0, sa0, sa1); issue(1, sb0, sb1); issue(2, sc0, sc1);
    if (PF == 4) issue(3, sd0, sd1);
    for (int sq = 0; sq <= r.nk; sq += PF) {
        step(sq, sa0, sa1);
        if (sq + 1 <= r.nk) step(sq + 1, sb0, sb1);
        if (sq + 2 <= r.nk) step(sq + 2, sc0, sc1);
        if (PF == 4 && sq + 3 <= r.nk) step(sq + 3, sd0, sd1);
    }
}

// Band apply of head g for every row of the tile: O_i[query c][d] = TAB_i[c][0][g] rows[<bos>][d] + sum over key rows / taps ...
// The rows of TWO list entries (32 keys) are staged and read back transposed once per chunk; every tile row that taps one of them
// runs its 4 MFMAs on the shared A fragments with its own banded coefficients (zeros for an entry it does not tap).
template <int ROWS, bool F16>
__device__ __forceinline__ void tile_band_apply(const S3Args& a, const TileM<ROWS>& r, const bf16_t* rows, int ldr, int g, const float* TAB,
                                                char* tile, f32x4 (&O)[ROWS][4]) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const int spb = r.c * r.TS + g;
    {   // <bos> slot
        const bf16_t* vb = rows + r.tok0 * ldr + g * DH + 4 * r.g4;
        uint2 u[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) u[db] = *reinterpret_cast<const uint2*>(vb + db * 16);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const float p0 = r.qok[i] ? TAB[i * r.WTS + spb] : 0.f;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                O[i][db] = f32x4{p0 * lo_t<F16>(u[db].x), p0 * hi_t<F16>(u[db].x), p0 * lo_t<F16>(u[db].y), p0 * hi_t<F16>(u[db].y)};
        }
    }
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const bf16_t* vbase = rows + r.tok0 * ldr + g * DH + gc * 8;
    int woff[4], sidx[4], troff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = vt_off(r8 + 8 * i, gc);
#pragma unroll
    for (int q = 0; q < 4; ++q) sidx[q] = spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH;
    {
        const int r0 = 4 * r.g4 + (r.c >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 16 + ((r.c & 3) << 2);
            troff[db] = vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8;
        }
    }
    uint4 st[4];
    auto fetch = [&](int pi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pj = pi + (i >> 1);
            const int tok = pj < r.nk ? r.ktok[pj] + r8 + 8 * (i & 1) : a.ntok;
            st[i] = ldg16_sel(vbase + (size_t)(tok < a.ntok ? tok : 0) * ldr, tok < a.ntok);
        }
    };
    fetch(0);
    for (int pi = 0; pi < r.nk; pi += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + woff[i]) = st[i];
        const int meta0 = r.kmeta[pi], meta1 = pi + 1 < r.nk ? r.kmeta[pi + 1] : -1;
        if (pi + 2 < r.nk) fetch(pi + 2);                                         // the next chunk's rows are in flight below
        const int ta0 = meta0 >> 8, m0 = (meta0 & 255) - 64, ta1 = meta1 >> 8, m1 = (meta1 & 255) - 64;
        __builtin_amdgcn_wave_barrier();                                          // LDS is in-order per wave: the tile is complete
        bf16x8 A[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db] + 2048));
            const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            A[db] = __builtin_bit_cast(bf16x8, v8);
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int tb0 = m0 - i + a.kh - 1, tb1 = m1 - i + a.kh - 1;
            const bool u0 = i < r.nrows && tb0 >= 0 && tb0 < a.kh, u1 = i < r.nrows && meta1 >= 0 && tb1 >= 0 && tb1 < a.kh;
            if (!(u0 || u1)) continue;                                            // (wave-uniform)
            const int jb0 = (1 + (ta0 * a.kh + (u0 ? tb0 : 0)) * a.kw) * NH, jb1 = u1 ? (1 + (ta1 * a.kh + tb1) * a.kw) * NH : 0;
            float pf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = r.tsel[j] >= 0 && r.qok[i];
                pf[j] = (on && u0) ? TAB[i * r.WTS + sidx[j] + jb0] : 0.f;
                pf[4 + j] = (on && u1) ? TAB[i * r.WTS + sidx[j] + jb1] : 0.f;
            }
            const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(pack2_t<F16>(pf[0], pf[1]), pack2_t<F16>(pf[2], pf[3]),
                                                                     pack2_t<F16>(pf[4], pf[5]), pack2_t<F16>(pf[6], pf[7])));
#pragma unroll
            for (int db = 0; db < 4; ++db) O[i][db] = mfma16<F16>(A[db], pb, O[i][db]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- round 6: the tile sweeps with their address arithmetic out of the key-row loop (see mfma_band_scores_fast): key-row list by v_readlane
// (nk <= 64), wave-uniform row base + 32-bit offsets, unmasked loads for key rows whose 16 tokens exist, one v_mad_u32_u24 per table address.
// No bias table (the training kernels).  Same loads, MFMAs, stores and values as tile_band_scores / tile_band_apply.
template <int ROWS, bool F16, int PF>
__device__ __forceinline__ void tile_band_scores_fast(const S3Args& a, const TileM<ROWS>& r, const bf16_t* rows, int ldr, const bf16_t* frag,
                                                      int ldf, int h, float* TAB, float mul, char* tile) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    bf16x8 qf0[ROWS], qf1[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const bf16_t* qrow = frag + (r.tok0 + (r.qok[i] ? r.iq[i] : 0)) * ldf + h * DH + r.g4 * 8;
        qf0[i] = __builtin_bit_cast(bf16x8, ldg16_sel(qrow, r.qok[i]));
        qf1[i] = __builtin_bit_cast(bf16x8, ldg16_sel(qrow + 32, r.qok[i]));
    }
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const char* sb = reinterpret_cast<const char*>(rows + r.tok0 * ldr + __builtin_amdgcn_readfirstlane(h) * DH);   // wave-uniform
    const unsigned ldb = (unsigned)ldr * 2u;
    const unsigned vb = (unsigned)gc * 16u, v0 = (unsigned)r8 * ldb + vb, v1 = v0 + 8u * ldb;
    const unsigned tab0 = lds_byte_addr(TAB);
    const int spb = r.c * r.TS + h;
    const int wpad = r.c * r.TS + r.J * NH + r.g4;                               // the lane's pad word of query c's table row
    unsigned wb4[ROWS][4], wm4[ROWS][4];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool on = r.tsel[q] >= 0 && r.qok[i];
            wb4[i][q] = tab0 + 4u * (unsigned)(i * r.WTS + (on ? spb + r.tsel[q] * NH : wpad));
            wm4[i][q] = on ? 4u * NH : 0u;
        }
    const int w0 = vt_off(r8, gc), w1 = vt_off(r8 + 8, gc);
    const int f0 = vt_off(r.c, r.g4), f1 = vt_off(r.c, 4 + r.g4);
    auto issue = [&](int sq, uint4& d0, uint4& d1) {
        if (sq > r.nk) return;                                                   // (past the end: the register set is never consumed)
        if (sq == 0) { d0 = ldg_u4(sb, vb); d1 = d0; return; }                   // sequence 0: every row is token 0 (<bos>)
        const int base = __builtin_amdgcn_readlane(r.vkt, sq - 1);
        const unsigned so = (unsigned)base * ldb;
        if (base + 16 <= a.ntok) { d0 = ldg_u4(sb, so + v0); d1 = ldg_u4(sb, so + v1); }
        else {
            d0 = base + r8 < a.ntok ? ldg_u4(sb, so + v0) : make_uint4(0, 0, 0, 0);
            d1 = base + r8 + 8 < a.ntok ? ldg_u4(sb, so + v1) : make_uint4(0, 0, 0, 0);
        }
    };
    auto step = [&](int sq, uint4& d0, uint4& d1) {
        *reinterpret_cast<uint4*>(tile + w0) = d0;
        *reinterpret_cast<uint4*>(tile + w1) = d1;
        issue(sq + PF, d0, d1);
        __builtin_amdgcn_wave_barrier();
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(tile + f0), k1 = *reinterpret_cast<const bf16x8*>(tile + f1);
        f32x4 sc[ROWS];
        if (sq == 0) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                if (i >= r.nrows) continue;
                sc[i] = mfma16<F16>(k1, qf1[i], mfma16<F16>(k0, qf0[i], f32x4{0.f, 0.f, 0.f, 0.f}));
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
                if (i < r.nrows && r.g4 == 0 && r.qok[i]) lds_st32(tab0 + 4u * (unsigned)(i * r.WTS + spb), sc[i][0] * mul);
        } else {
            const int meta = __builtin_amdgcn_readlane(r.vkm, sq - 1), ta = meta >> 8, m = (meta & 255) - 64;
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int tb = m - i + a.kh - 1;
                if (i >= r.nrows || tb < 0 || tb >= a.kh) continue;              // (wave-uniform)
                sc[i] = mfma16<F16>(k1, qf1[i], mfma16<F16>(k0, qf0[i], f32x4{0.f, 0.f, 0.f, 0.f}));
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int tb = m - i + a.kh - 1;
                if (i >= r.nrows || tb < 0 || tb >= a.kh) continue;
                const unsigned jb = (unsigned)(1 + (ta * a.kh + tb) * a.kw);
#pragma unroll
                for (int q = 0; q < 4; ++q) lds_st32(__umul24(jb, wm4[i][q]) + wb4[i][q], sc[i][q] * mul);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    static_assert(PF == 3 || PF == 4, "named register sets");
    uint4 sa0 = make_uint4(0, 0, 0, 0), sa1 = sa0, sb0 = sa0, sb1 = sa0, sc0 = sa0, sc1 = sa0, sd0 = sa0, sd1 = sa0;
    issue(0, sa0, sa1); issue(1, sb0, sb1); issue(2, sc0, sc1);
    if (PF == 4) issue(3, sd0, sd1);
    for (int sq = 0; sq <= r.nk; sq += PF) {
        step(sq, sa0, sa1);
        if (sq + 1 <= r.nk) step(sq + 1, sb0, sb1);
        if (sq + 2 <= r.nk) step(sq + 2, sc0, sc1);
        if (PF == 4 && sq + 3 <= r.nk) step(sq + 3, sd0, sd1);
    }
}

template <int ROWS, bool F16>
__device__ __forceinline__ void tile_band_apply_fast(const S3Args& a, const TileM<ROWS>& r, const bf16_t* rows, int ldr, int g, const float* TAB,
                                                     char* tile, f32x4 (&O)[ROWS][4]) {
    constexpr int NH = S3M_NH, DH = S3M_DH;
    const int spb = r.c * r.TS + g;
    {   // <bos> slot
        const bf16_t* vb = rows + r.tok0 * ldr + g * DH + 4 * r.g4;
        uint2 u[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) u[db] = *reinterpret_cast<const uint2*>(vb + db * 16);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const float p0 = r.qok[i] ? TAB[i * r.WTS + spb] : 0.f;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                O[i][db] = f32x4{p0 * lo_t<F16>(u[db].x), p0 * hi_t<F16>(u[db].x), p0 * lo_t<F16>(u[db].y), p0 * hi_t<F16>(u[db].y)};
        }
    }
    const int gc = r.lane & 7, r8 = r.lane >> 3;
    const char* sb = reinterpret_cast<const char*>(rows + r.tok0 * ldr + __builtin_amdgcn_readfirstlane(g) * DH);   // wave-uniform
    const unsigned ldb = (unsigned)ldr * 2u;
    const unsigned v0 = (unsigned)r8 * ldb + (unsigned)gc * 16u, v1 = v0 + 8u * ldb;
    const unsigned tab0 = lds_byte_addr(TAB);
    int woff[4], troff[4];
    unsigned sidx4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = vt_off(r8 + 8 * i, gc);
#pragma unroll
    for (int q = 0; q < 4; ++q) sidx4[q] = tab0 + 4u * (unsigned)(spb + (r.tsel[q] < 0 ? 0 : r.tsel[q]) * NH);
    {
        const int r0 = 4 * r.g4 + (r.c >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 16 + ((r.c & 3) << 2);
            troff[db] = vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8;
        }
    }
    uint4 st[4];
    auto fetch2 = [&](int pj, uint4& d0, uint4& d1) {
        if (pj >= r.nk) { d0 = d1 = make_uint4(0, 0, 0, 0); return; }           // (the odd entry of the last pair: its rows must be zeros)
        const int base = __builtin_amdgcn_readlane(r.vkt, pj);
        const unsigned so = (unsigned)base * ldb;
        if (base + 16 <= a.ntok) { d0 = ldg_u4(sb, so + v0); d1 = ldg_u4(sb, so + v1); }
        else {
            d0 = base + r8 < a.ntok ? ldg_u4(sb, so + v0) : make_uint4(0, 0, 0, 0);
            d1 = base + r8 + 8 < a.ntok ? ldg_u4(sb, so + v1) : make_uint4(0, 0, 0, 0);
        }
    };
    fetch2(0, st[0], st[1]); fetch2(1, st[2], st[3]);
    for (int pi = 0; pi < r.nk; pi += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + woff[i]) = st[i];
        const int meta0 = __builtin_amdgcn_readlane(r.vkm, pi), meta1 = pi + 1 < r.nk ? __builtin_amdgcn_readlane(r.vkm, pi + 1) : -1;
        if (pi + 2 < r.nk) { fetch2(pi + 2, st[0], st[1]); fetch2(pi + 3, st[2], st[3]); }
        const int ta0 = meta0 >> 8, m0 = (meta0 & 255) - 64, ta1 = meta1 >> 8, m1 = (meta1 & 255) - 64;
        __builtin_amdgcn_wave_barrier();
        bf16x8 A[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + troff[db] + 2048));
            const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            A[db] = __builtin_bit_cast(bf16x8, v8);
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int tb0 = m0 - i + a.kh - 1, tb1 = m1 - i + a.kh - 1;
            const bool u0 = i < r.nrows && tb0 >= 0 && tb0 < a.kh, u1 = i < r.nrows && meta1 >= 0 && tb1 >= 0 && tb1 < a.kh;
            if (!(u0 || u1)) continue;                                            // (wave-uniform)
            const unsigned rb = 4u * (unsigned)(i * r.WTS);
            const unsigned jb0 = rb + 4u * NH * (unsigned)(1 + (ta0 * a.kh + (u0 ? tb0 : 0)) * a.kw);
            const unsigned jb1 = u1 ? rb + 4u * NH * (unsigned)(1 + (ta1 * a.kh + tb1) * a.kw) : jb0;
            float pf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = r.tsel[j] >= 0 && r.qok[i];
                const float x0 = lds_ld32(sidx4[j] + jb0), x1 = lds_ld32(sidx4[j] + jb1);
                pf[j] = (on && u0) ? x0 : 0.f;
                pf[4 + j] = (on && u1) ? x1 : 0.f;
            }
            const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(pack2_t<F16>(pf[0], pf[1]), pack2_t<F16>(pf[2], pf[3]),
                                                                     pack2_t<F16>(pf[4], pf[5]), pack2_t<F16>(pf[6], pf[7])));
#pragma unroll
            for (int db = 0; db < 4; ++db) O[i][db] = mfma16<F16>(A[db], pb, O[i][db]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int ROWS, bool F16, bool BIAS>      // BIAS: a.bias != NULL (the relative-position bias of cfg 5; compiled out of the training kernels otherwise)
__global__ __launch_bounds__(512, ROWS >= 4 ? 1 : 2) void s3_fwd_tile_kernel(S3Args a) {
    constexpr int NH = S3M_NH, DH = S3M_DH, W = S3M_W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int J = a.kf * a.kh * a.kw + 1, WTS = W * s3m_ts(J);
    float* SP = reinterpret_cast<float*>(smem);                                  // [ROWS][W][J * NH + 4]
    char* vt_base = smem + (size_t)ROWS * WTS * sizeof(float);                   // 8 wave-private [32][64] bf16 tiles
    __shared__ float wsh[64];
    __shared__ int ktok[S3T_MAXK], kmeta[S3T_MAXK], kcnt[1];
    const int t = threadIdx.x;
    const int tiles = a.F * (a.H / ROWS);
    const int bid = xcd_row_id();
    const int b = bid / tiles, t2 = bid % tiles;
    int f, y0;
    s3t_tile_order<ROWS>(a, t2, f, y0);
    const int ry0 = f * a.H + y0;
    if (t < NH * NH) wsh[t] = a.wth[t];
    if (ry0 == 0) {                                                              // <bos> output row = its own value
        for (int e = t; e < NH * DH; e += blockDim.x) {
            const bf16_t raw = a.v[((size_t)b * a.ntok) * a.ld + e];
            if (F16) {
                bf16_t hi, lo;
                f2bf_hilo((float)__builtin_bit_cast(_Float16, raw), hi, lo);
                if (a.o) a.o[((size_t)b * a.ntok) * a.ldo + e] = hi;     // (o == NULL: the fp16 copy alone -- the fp16-gradient backward reads nothing else)
                if (a.ol) a.ol[((size_t)b * a.ntok) * a.ldo + e] = a.ol_f16 ? raw : lo;
            } else a.o[((size_t)b * a.ntok) * a.ldo + e] = raw;
        }
    }
    int nrows = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) nrows += ((ry0 + i * a.dh) * W + 1 < a.ntok) ? 1 : 0;   // (rows of a tile exist in order)
    if (nrows == 0) return;                                                      // whole tile beyond the sequence (uniform)
    {   // (WTS is a multiple of 4 floats: 16-byte stores, a quarter of the ds_write instructions)
        const float4 neg4 = make_float4(NEG_MAX, NEG_MAX, NEG_MAX, NEG_MAX);
        for (int e = t; e < nrows * WTS / 4; e += blockDim.x) reinterpret_cast<float4*>(SP)[e] = neg4;
    }
    s3t_keylist<ROWS>(a, f, y0, ktok, kmeta, kcnt);
    const TileM<ROWS> r = s3t_init<ROWS>(a, b, f, y0, nrows, ktok, kmeta, kcnt[0]);
    if (!(a.dbg & 1))     // (a.dbg, tuning key 9: bits 0 / 1 / 2 skip the score / softmax + mix / apply phase -- timing probes, garbage results)
    {
        if (!BIAS && r.nk <= 64) tile_band_scores_fast<ROWS, F16, ROWS >= 4 ? 4 : S3M_PF>(a, r, a.k, a.ld, a.q, a.ld, r.wave, SP, 1.f, vt_base + r.wave * 4096);
        else tile_band_scores<ROWS, F16, ROWS >= 4 ? 4 : S3M_PF>(a, r, a.k, a.ld, a.q, a.ld, r.wave, SP, BIAS ? a.scale : 1.f, BIAS ? a.bias : nullptr, vt_base + r.wave * 4096);
    }
    __syncthreads();
    for (int i = 0; i < nrows && !(a.dbg & 2); ++i) rowm_softmax(SP + i * WTS, J, nullptr, 0, BIAS ? 1.f : a.scale);
    __syncthreads();
    // talking heads: P'[g] = sum_h Wth[g][h] P[h] per (row, w, j), in place
    float wr[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wr[k] = wsh[k];
    const float rJ = 1.f / (float)J, rWJ = 1.f / (float)(W * J);
    for (int it = t; it < nrows * W * J && !(a.dbg & 2); it += blockDim.x) {
        const int i = (int)(((float)it + 0.5f) * rWJ), item = it - i * W * J;
        float pv[8], out[8];
        const int ib = i * WTS + s3m_item(item, rJ);
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) pv[hh] = SP[ib + hh];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float s = 0.f;
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) s += wr[g * NH + hh] * pv[hh];
            S3_MIX_PIN_ASM(s);                   // (one head at a time: see lds_store8_done)
            out[g] = s;
        }
        lds_store8_done(SP + ib, out);
    }
    __syncthreads();
    {
        const int g = r.wave;
        f32x4 O[ROWS][4];
        if (!(a.dbg & 4)) {
            if (r.nk <= 64) tile_band_apply_fast<ROWS, F16>(a, r, a.v, a.ld, g, SP, vt_base + r.wave * 4096, O);
            else tile_band_apply<ROWS, F16>(a, r, a.v, a.ld, g, SP, vt_base + r.wave * 4096, O);
        }
        else {
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
#pragma unroll
                for (int db = 0; db < 4; ++db) O[i][db] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if (!r.qok[i]) continue;
            const size_t go = (r.tok0 + r.iq[i]) * a.ldo + g * DH + 4 * r.g4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const uint32_t h01 = pack2_rne(O[i][db][0], O[i][db][1]), h23 = pack2_rne(O[i][db][2], O[i][db][3]);
                if (!F16 || a.o) *reinterpret_cast<uint2*>(a.o + go + db * 16) = make_uint2(h01, h23);
                if (F16 && a.ol)
                    *reinterpret_cast<uint2*>(a.ol + go + db * 16) = a.ol_f16 ?
                        make_uint2(pack2_f16_sat(O[i][db][0], O[i][db][1]), pack2_f16_sat(O[i][db][2], O[i][db][3])) :
                        make_uint2(pack2_rne(O[i][db][0] - lo_f(h01), O[i][db][1] - hi_f(h01)), pack2_rne(O[i][db][2] - lo_f(h23), O[i][db][3] - hi_f(h23)));
            }
        }
    }
}

// MFMA backward, query side (same row / wave mapping as the forward): recompute P, dP' = dO . V^T (band scores with dO as the
// fragment and V as the rows), dW_th partial, dP = W^T dP', ds = P (dP - sum P dP), dq = scale * ds . K (band apply over K);
// ds and P' go to the fp32 workspace for the key-side kernel, the <bos> key / value partials to part_k0 / part_v0.
// LDS: R1 = SP (P) until ds exists, then the 8 transposed K tiles | DP | RED [8][64] | PM0 [W][NH]
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr float S3Q_DPS = 0.015625f;      // factor on dP' (and everything derived from it up to ds) in the fp16-gradient form: see the item pass
template <bool BIAS, bool G16 = false>     // G16: the fp16-gradient form (S3Args::gs2)
__global__ __launch_bounds__(512, 4) void s3_bwd_q_mfma_kernel(S3Args a) {      // (4 waves per SIMD = two workgroups per CU: <= 128 registers)
    constexpr int NH = S3M_NH, DH = S3M_DH, W = S3M_W, inner = NH * DH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int J = a.kf * a.kh * a.kw + 1, TS = s3m_ts(J), nsp = W * TS;
    const float rJ = 1.f / (float)J;
    const size_t r1 = (size_t)nsp * 4 > 8 * 4096 ? (size_t)nsp * 4 : 8 * 4096;
    float* SP = reinterpret_cast<float*>(smem);
    float* DP = reinterpret_cast<float*>(smem + r1);
    float* RED = DP + nsp;                                                       // [8][64]
    float* PM0 = RED + 8 * 64;                                                   // [W][NH]  P' of the <bos> slot
    __shared__ float wsh[64];
    __shared__ int pslot[S3M_PLANES + 1], ptok[S3M_PLANES + 1];
    const int t = threadIdx.x;
    const int rows = a.F * a.H;
    const int bid = xcd_row_id();
    const int b = bid / rows, r2 = bid % rows;
    int f, y;
    s3m_row_order(a, r2, f, y);
    const int ry = f * a.H + y;
    const int nq = a.ntok - 1;
    if (t < NH * NH) wsh[t] = a.wth[t];
    float* pth = a.part_th + (size_t)bid * NH * NH;
    float* pk0 = a.part_k0 + (size_t)bid * inner;
    float* pv0 = a.part_v0 + (size_t)bid * inner;
    if (ry == 0)   // dq of the <bos> row is zero (its query is never used)
        for (int e = t; e < inner; e += blockDim.x) a.dq[((size_t)b * a.ntok) * a.ldd + e] = 0;
    if (ry * W + 1 >= a.ntok) {   // row beyond the sequence: contributes nothing
        for (int e = t; e < NH * NH; e += blockDim.x) pth[e] = 0.f;
        for (int e = t; e < inner; e += blockDim.x) { pk0[e] = 0.f; pv0[e] = 0.f; }
        return;
    }
    {   // (nsp is a multiple of 4 floats and both tables start 16-byte aligned: 16-byte stores)
        const float4 neg4 = make_float4(NEG_MAX, NEG_MAX, NEG_MAX, NEG_MAX), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = t; e < nsp / 4 && !(a.dbg & 2048); e += blockDim.x) { reinterpret_cast<float4*>(SP)[e] = neg4; reinterpret_cast<float4*>(DP)[e] = zero4; }
    }
    rowm_planes(a, f, y, pslot, ptok);
    const RowM r = rowm_init(a, b, ry, pslot, ptok);
    char* stile = reinterpret_cast<char*>(PM0 + W * NH) + r.wave * 2048;                    // 8 wave-private [16][64] bf16 staging tiles
    // (a.dbg, tuning key 17: timing probes only -- bit 0 skips the two score sweeps, bit 1 the ds / P' workspace stores, bit 2 the dq
    //  apply sweep, bit 3 the dW_th partial; results are garbage)
    if (!(a.dbg & 1)) {
    if constexpr (!BIAS) {
    mfma_band_scores_fast<G16>(a, r, a.k, a.ld, a.q, a.ld, r.wave, SP, 1.f, stile);                   // scores (raw products: see rowm_softmax)
    mfma_band_scores_fast<G16>(a, r, a.v, a.ld, a.dO, a.lddo, r.wave, DP, G16 ? S3Q_DPS : 1.f, stile); // dP'[g] = dO[g] . v_j[g]  (G16: times S * 2^-6, as everything derived from it up to ds)
    } else {
    mfma_band_scores_staged<G16>(a, r, a.k, a.ld, a.q, a.ld, r.wave, SP, BIAS ? a.scale : 1.f, BIAS ? a.bias : nullptr, stile);   // scores (raw products unless a bias table is added: see rowm_softmax)
    mfma_band_scores_staged<G16>(a, r, a.v, a.ld, a.dO, a.lddo, r.wave, DP, G16 ? S3Q_DPS : 1.f, nullptr, stile);   // dP'[g] = dO[g] . v_j[g]
    }
    }
    __syncthreads();
    // recomputing key side: this kernel leaves (row max, 1 / row sum, delta) per (query, head) instead of the ds / P' workspace
    float* gst = a.stats ? a.stats + ((size_t)b * nq + (size_t)ry * W) * NH * 4 : nullptr;
    const int wvalid = a.ntok - 1 - ry * W;                                                 // queries of this row that exist
    if (!(a.dbg & 64)) rowm_softmax(SP, J, gst, wvalid, BIAS ? 1.f : a.scale);             // P   (more probe bits of key 17: 64 no softmax, 128 no item pass, 256 no ds pass, 512 no pack pass, 1024 no <bos> partials, 2048 no table init)
    __syncthreads();
    // ONE pass over the (query, slot) items, all 8 heads of an item in the thread's registers (tuning key 19 = 1 restores the three
    // separate passes it replaces):
    //   P'[g]  = sum_h W[g][h] P[h]          -> global (the key side needs it; two 16-byte stores), the <bos> slot also to PM0
    //   dW_th[g][h] += dP'[g] P[h]           -> 64 per-thread partial sums, reduced over the workgroup in a fixed order below
    //   dP[h]  = sum_g W[g][h] dP'[g]        -> in place of dP'
    // The separate dW_th pass (thread = one (g, h) pair walking all 736 items: 2 LDS reads + ~8 index instructions per FMA) was the
    // largest block of this kernel (probe: -224 us of 1838 with it skipped, tools/attn_probe.py); here it costs 64 FMAs per item on
    // values that are in registers anyway.  Absent queries hold dP' = 0 (their dO fragment is zero) and add nothing.
    if constexpr (G16) {
        // Round 6, the item pass on the MATRIX pipe (fp16-gradient form only: there dP' carries the gradient scale S and sits in fp16's range; the
        // band sweep above wrote it times 2^-6 -- 64 products of |S dO| <= 2^10 -- and ds below takes the factor back).  A wave takes 32 items
        // (query, slot) at a time as the B operand of v_mfma_f32_16x16x16_f16: lane (n, kg) holds heads 4 (kg & 1) .. + 3 of item 16 (kg >> 1) + n,
        // one ds_read_b128 of the table.  The A operand is the 8 x 8 mix matrix twice on the diagonal of a 16 x 16 block (rows / k 0..7: the
        // first 16 items, 8..15: the second 16), as an fp16 hi + lo pair so that only dP' (and P in the pack pass) is rounded:
        //     D[h | 8 + h][item] = sum_g W[g][h] dP'[g][item]      -- and lane (n, kg) of D is again (item, heads 4 (kg & 1) .. + 3): stored in place.
        // dW_th[g][h] = sum over the items of dP'[g] P[h] contracts over the ITEMS: A = dP' of head (m & 7) for 4 items, B = P of head (n & 7)
        // for the same 4 items (scalar table reads, their order rotated per lane group so that the 32 lanes of an LDS pass hit 32 banks); rows /
        // columns 0..7 run on the first 16 items of the group and 8..15 on the second, the two diagonal blocks of the accumulator are the two
        // partial sums.  Against the VALU pass it replaces: 192 FMAs and 48 LDS reads of W per item -> 5 MFMAs per 32 items.
        const int lane = t & 63, wv = t >> 6, n16 = lane & 15, kg = lane >> 4, sset = n16 >> 3, kh = kg & 1;
        const bool on = (n16 < 8) == (kg < 2);
        f16x4_t awt_hi, awt_lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float wt = on ? wsh[(4 * kh + i) * NH + (n16 & 7)] : 0.f;          // A[r = h][k = g] = W[g][h]
            const _Float16 h16 = (_Float16)wt;
            awt_hi[i] = h16; awt_lo[i] = (_Float16)(wt - (float)h16);
        }
        const int nitems = W * J, ngrp = (nitems + 31) >> 5;
        f32x4 CW = {0.f, 0.f, 0.f, 0.f};
        float dp_amax = 0.f;
        for (int G = wv; G < ngrp && !(a.dbg & 128); G += 8) {
            {   // dW_th operands first: this group's dP' is overwritten below
                const int ib0 = 32 * G + 16 * sset + 4 * kg;
                float avf[4], bvf[4];
#pragma unroll
                for (int ts = 0; ts < 4; ++ts) {
                    const int item = ib0 + ((ts + sset + 2 * kh) & 3);
                    const bool ok = item < nitems;
                    const int idx = s3m_item(ok ? item : 0, rJ) + (n16 & 7);
                    avf[ts] = ok ? DP[idx] : 0.f;
                    bvf[ts] = ok ? SP[idx] : 0.f;
                }
                const f16x4_t av = __builtin_bit_cast(f16x4_t, make_uint2(pack2_f16_sat(avf[0], avf[1]), pack2_f16_sat(avf[2], avf[3])));
                const f16x4_t bv = __builtin_bit_cast(f16x4_t, make_uint2(pack2_f16(bvf[0], bvf[1]), pack2_f16(bvf[2], bvf[3])));
                CW = __builtin_amdgcn_mfma_f32_16x16x16f16(av, bv, CW, 0, 0, 0);
            }
            const int item = 32 * G + 16 * (kg >> 1) + n16;
            const bool ok = item < nitems;
            const int ib = s3m_item(ok ? item : 0, rJ) + 4 * kh;
            const float4 d4 = *reinterpret_cast<const float4*>(DP + ib);
            // (dP' x 2^-6 leaves for fp16 through the saturating, counted converter: every table entry passes here exactly once)
            const f16x4_t df = __builtin_bit_cast(f16x4_t, make_uint2(pack2_f16_sat_n(d4.x, d4.y, dp_amax), pack2_f16_sat_n(d4.z, d4.w, dp_amax)));
            f32x4 rr = {0.f, 0.f, 0.f, 0.f};
            rr = __builtin_amdgcn_mfma_f32_16x16x16f16(awt_hi, df, rr, 0, 0, 0);
            rr = __builtin_amdgcn_mfma_f32_16x16x16f16(awt_lo, df, rr, 0, 0, 0);
            if (ok) *reinterpret_cast<float4*>(DP + ib) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        }
        f16_sat_commit(dp_amax);
        // the wave's dW_th: diagonal block (g, h) of lane (h, kg < 2) + block (8 + g, 8 + h) of lane (8 + h, kg + 2) = lane + 40
        float tw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[i] = CW[i] + __shfl(CW[i], (lane + 40) & 63, 64);
        if (n16 < 8 && kg < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) RED[wv * 64 + (4 * kg + i) * NH + n16] = tw[i] * (1.f / S3Q_DPS);
        }
        __syncthreads();
        if (t < NH * NH) {
            float sum = 0.f;
            for (int k = 0; k < 8; ++k) sum += RED[k * 64 + t];
            pth[t] = sum;
        }
        __syncthreads();
    } else if (!a.sep_passes) {
        // two sweeps over the items so that only 32 of the 64 dW_th partial sums are live at a time (with all 64 next to the mix
        // operands the compiler spills ~130 registers at the 128-register budget of two workgroups per CU):
        //   sweep 0: P' (-> global, PM0) and dW_th rows g = 0..3;   sweep 1: dW_th rows g = 4..7 and dP (in place of dP')
        float red_lane = 0.f;                      // this lane's entry of the wave's 64 sums (entry index = lane)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float acc[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] = 0.f;
#pragma unroll 1
            for (int item = t; item < W * J && !(a.dbg & 128); item += blockDim.x) {
                const int wq = (int)(((float)item + 0.5f) * rJ), j = item - wq * J;
                const int iq = 1 + ry * W + wq;
                const int ib = item * NH + wq * S3M_PAD;
                const float4 p0 = *reinterpret_cast<const float4*>(SP + ib), p1 = *reinterpret_cast<const float4*>(SP + ib + 4);
                const float4 d0 = *reinterpret_cast<const float4*>(DP + ib), d1 = *reinterpret_cast<const float4*>(DP + ib + 4);
                const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                const float dv_[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                const float* wv = wsh;                 // (opaque per iteration: hoisted out of the loop the 64 values of W cost registers)
                asm volatile("" : "+v"(wv));
                float res[8];                          // sweep 0: P'[g];  sweep 1: dP[h]
#pragma unroll
                for (int x = 0; x < 8; ++x) res[x] = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float4 w0 = *reinterpret_cast<const float4*>(wv + g * NH), w1 = *reinterpret_cast<const float4*>(wv + g * NH + 4);
                    const float wg[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                    if (half == 0) {
                        if (!a.packed) {                   // (packed workspace: P' is mixed in the pack loop further down, next to ds)
                            float sm = 0.f;
#pragma unroll
                            for (int hh = 0; hh < 8; ++hh) sm += wg[hh] * pv[hh];
                            res[g] = sm;
                        }
                    } else {
#pragma unroll
                        for (int hh = 0; hh < 8; ++hh) res[hh] += wg[hh] * dv_[g];
                    }
                    if ((g >> 2) == half && !(a.dbg & 8)) {
#pragma unroll
                        for (int hh = 0; hh < 8; ++hh) acc[(g & 3) * NH + hh] = fmaf(dv_[g], pv[hh], acc[(g & 3) * NH + hh]);
                    }
                }
                if (half == 0) {
                    if (!a.packed) {
                        if (iq < a.ntok && !gst && !(a.dbg & 2)) {
                            float4* dst = reinterpret_cast<float4*>(a.pm + (((size_t)b * nq + (iq - 1)) * J + j) * NH);
                            dst[0] = make_float4(res[0], res[1], res[2], res[3]);
                            dst[1] = make_float4(res[4], res[5], res[6], res[7]);
                        }
                        if (j == 0) {
#pragma unroll
                            for (int g = 0; g < 8; ++g) PM0[wq * NH + g] = iq < a.ntok ? res[g] : 0.f;
                        }
                    }
                } else {
                    *reinterpret_cast<float4*>(DP + ib) = make_float4(res[0], res[1], res[2], res[3]);
                    *reinterpret_cast<float4*>(DP + ib + 4) = make_float4(res[4], res[5], res[6], res[7]);
                }
            }
            // wave sums of the 32 partials of this half, fixed order: halving butterfly; lane l (< 32) ends with entry l of the half
            const int lane = t & 63;
#pragma unroll
            for (int off = 16, cnt = 16; off >= 1; off >>= 1, cnt >>= 1) {
                const bool upper = (lane & off) != 0;
#pragma unroll
                for (int i = 0; i < cnt; ++i) {
                    const float send = upper ? acc[i] : acc[i + cnt];
                    const float keep = upper ? acc[i + cnt] : acc[i];
                    acc[i] = keep + __shfl_xor(send, off, 64);
                }
            }
            const float tot = acc[0] + __shfl_xor(acc[0], 32, 64);       // lanes l and l + 32 hold the two halves of entry l & 31
            if ((lane >> 5) == half) red_lane = tot;                     // lane = 32 * half + entry: the wave's sum of dW_th[4 half + e / 8][e % 8]
        }
        RED[(t >> 6) * 64 + (t & 63)] = red_lane;
        __syncthreads();
        if (t < NH * NH) {
            float s = 0.f;
            for (int k = 0; k < 8; ++k) s += RED[k * 64 + t];
            pth[t] = s;
        }
        __syncthreads();
    } else {
        float wr[64];                              // the 8 x 8 mix matrix in registers for the loops below
#pragma unroll
        for (int k = 0; k < 64; ++k) wr[k] = wsh[k];
        // P' = mix(P) -> global (the key side needs it), P stays in SP;  P' of the <bos> slot also to PM0
        for (int item0 = t; item0 < (gst ? W : W * J); item0 += blockDim.x) {           // (recomputing key side: only the <bos> slots are needed)
            const int item = gst ? item0 * J : item0;
            const int wq = (int)(((float)item + 0.5f) * rJ), j = item - wq * J;
            const int iq = 1 + ry * W + wq;
            float pv[8];
    #pragma unroll
            for (int hh = 0; hh < 8; ++hh) pv[hh] = SP[item * NH + wq * S3M_PAD + hh];
            float* dst = a.pm + (((size_t)b * nq + (iq - 1)) * J + j) * NH;
    #pragma unroll
            for (int g = 0; g < 8; ++g) {
                float s = 0.f;
    #pragma unroll
                for (int hh = 0; hh < 8; ++hh) s += wr[g * NH + hh] * pv[hh];
                if (iq < a.ntok && !gst && !(a.dbg & 2)) dst[g] = s;
                if (j == 0) PM0[wq * NH + g] = iq < a.ntok ? s : 0.f;
            }
        }
        // dW_th[g][h] partial = sum_{w,j} dP'[g] * P[h]   (thread = (g,h) pair x 8 item groups).  Rows of absent queries hold
        // dP' = 0 (their dO fragment is zero), so they add nothing.
        {
            const int pair = t & 63, grp = t >> 6;
            const int g = pair / NH, hh = pair % NH;
            float acc = 0.f;
            for (int item = grp; item < W * J && !(a.dbg & 8); item += 8 * 8) {       // eight (dP', P) pairs in flight, added in order (was one dependent LDS pair per iteration, 92 of them)
                float dv8[8], pv8[8];
    #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = item + 8 * u, itc = it < W * J ? it : grp;
                    const int ib = s3m_item(itc, rJ);
                    dv8[u] = DP[ib + g]; pv8[u] = SP[ib + hh];
                }
    #pragma unroll
                for (int u = 0; u < 8; ++u) acc += item + 8 * u < W * J ? dv8[u] * pv8[u] : 0.f;
            }
            RED[grp * 64 + pair] = acc;
            __syncthreads();
            if (t < NH * NH) {
                float s = 0.f;
                for (int k = 0; k < 8; ++k) s += RED[k * 64 + t];
                pth[t] = s;
            }
            __syncthreads();
        }
        // dP[h] = sum_g Wth[g][h] dP'[g]   (in place, item-local)
        for (int item = t; item < W * J; item += blockDim.x) {
            float dv_[8], out[8];
            const int ib = s3m_item(item, rJ);
    #pragma unroll
            for (int g = 0; g < 8; ++g) dv_[g] = DP[ib + g];
    #pragma unroll
            for (int hh = 0; hh < 8; ++hh) {
                float s = 0.f;
    #pragma unroll
                for (int g = 0; g < 8; ++g) s += wr[g * NH + hh] * dv_[g];
                S3_MIX_PIN_ASM(s);               // (one head at a time: see lds_store8_done)
                out[hh] = s;
            }
            lds_store8_done(DP + ib, out);
        }
        __syncthreads();
    }
    // ds = P * (dP - sum_j P dP)  -> DP and the global workspace
    {
        const int cc = t & 3, wh = t >> 2, h = wh % NH, w = wh / NH;
        const int i = 1 + ry * W + w;
        float ds_amax = 0.f;                                      // G16: S ds is clamped to the fp16 range HERE (every later use packs it to fp16) and counted
        if (a.dbg & 256) {
        } else if (J <= 48) {                                            // the thread's <= 12 slots in registers (see rowm_softmax); same order of operations
            float pv[12], dv[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int j = cc + 4 * k, idx = w * TS + (j < J ? j : cc) * NH + h;
                pv[k] = SP[idx]; dv[k] = DP[idx];
            }
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < 12; ++k) if (cc + 4 * k < J) d += pv[k] * dv[k];
            d = quad_sum(d);
            if (gst && cc == 0 && i < a.ntok) gst[(w * NH + h) * 4 + 2] = d;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int j = cc + 4 * k;
                if (j < J) {
                    const int idx = w * TS + j * NH + h;
                    float dsv = pv[k] * (dv[k] - d);
                    if constexpr (G16) { dsv *= 1.f / S3Q_DPS; ds_amax = fmaxf(ds_amax, fabsf(dsv)); dsv = f16_clamp(dsv); }
                    DP[idx] = dsv;
                    if (i < a.ntok && !gst && !a.packed && !(a.dbg & 2)) a.ds[(((size_t)b * nq + (i - 1)) * J + j) * NH + h] = dsv;
                }
            }
        } else {
        float d = 0.f;
        for (int j = cc; j < J; j += 4) d += SP[w * TS + j * NH + h] * DP[w * TS + j * NH + h];
        d = quad_sum(d);
        if (gst && cc == 0 && i < a.ntok) gst[(w * NH + h) * 4 + 2] = d;
        for (int j = cc; j < J; j += 4) {
            const int idx = w * TS + j * NH + h;
            float dsv = SP[idx] * (DP[idx] - d);
            if constexpr (G16) { dsv *= 1.f / S3Q_DPS; ds_amax = fmaxf(ds_amax, fabsf(dsv)); dsv = f16_clamp(dsv); }
            DP[idx] = dsv;
            if (i < a.ntok && !gst && !a.packed) a.ds[(((size_t)b * nq + (i - 1)) * J + j) * NH + h] = dsv;
        }
        }
        if constexpr (G16) f16_sat_commit(ds_amax);
    }
    __syncthreads();
    // Packed workspace (round 5): ONE pass over the items writes (bf16 ds | bf16 P') words, 8 heads = two 16-byte stores per (query, slot) -- the
    // fp32 form wrote P' as two 16-byte stores in the item pass above and ds as twelve scattered 4-byte stores per thread in the ds pass: 1.98 GB
    // per call at b = 128, read back by the key side with two 4-byte loads per coefficient.  Same values as before: the key side rounded both to
    // bf16 (round to nearest even) for its MFMA operands anyway, so dK / dV do not change by a bit.  P' is mixed here (the item pass skips it).
    if constexpr (G16) {
        // pack pass of the fp16-gradient form on the matrix pipe: P' = W P for 32 items per MFMA pair (same operand shape as the item pass above),
        // lane (n, kg) ends with P'[4 (kg & 1) .. + 3] of its item next to the ds it reads from the table: four (fp16 S ds | fp16 P') words, ONE
        // 16-byte store.  P' of the <bos> slot also goes to PM0.  Entries the key side never reads are not written (see the VALU form below).
        const int lane = t & 63, wv = t >> 6, n16 = lane & 15, kg = lane >> 4, kh = kg & 1;
        const bool on = (n16 < 8) == (kg < 2);
        f16x4_t aw_hi, aw_lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ww = on ? wsh[(n16 & 7) * NH + 4 * kh + i] : 0.f;            // A[r = g][k = h] = W[g][h]
            const _Float16 h16 = (_Float16)ww;
            aw_hi[i] = h16; aw_lo[i] = (_Float16)(ww - (float)h16);
        }
        const int ta0 = max(0, a.kf - 1 - f / a.df), tb0 = max(0, a.kh - 1 - y / a.dh);
        const float rkw = 1.f / (float)a.kw, rkh = 1.f / (float)a.kh;
        const int nitems = W * J, ngrp = (nitems + 31) >> 5;
        for (int G = wv; G < ngrp && !(a.dbg & 512); G += 8) {
            const int item = 32 * G + 16 * (kg >> 1) + n16;
            const bool ok = item < nitems;
            const int itc = ok ? item : 0;
            const int wq = (int)(((float)itc + 0.5f) * rJ), j = itc - wq * J;
            const int iq = 1 + ry * W + wq;
            bool keep = false;
            if (j > 0) {
                const int pl = (int)(((float)(j - 1) + 0.5f) * rkw), tc = j - 1 - pl * a.kw;
                const int ta = (int)(((float)pl + 0.5f) * rkh), tb = pl - ta * a.kh;
                keep = ta >= ta0 && tb >= tb0 && wq - (a.kw - 1 - tc) * a.dw >= 0;
            }
            const int ib = itc * NH + wq * S3M_PAD + 4 * kh;
            const float4 p4 = *reinterpret_cast<const float4*>(SP + ib);
            const float4 s4 = *reinterpret_cast<const float4*>(DP + ib);
            const f16x4_t pf = __builtin_bit_cast(f16x4_t, ok ? make_uint2(pack2_f16(p4.x, p4.y), pack2_f16(p4.z, p4.w)) : make_uint2(0u, 0u));
            f32x4 pm = {0.f, 0.f, 0.f, 0.f};
            pm = __builtin_amdgcn_mfma_f32_16x16x16f16(aw_hi, pf, pm, 0, 0, 0);
            pm = __builtin_amdgcn_mfma_f32_16x16x16f16(aw_lo, pf, pm, 0, 0, 0);
            if (ok && j == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) PM0[wq * NH + 4 * kh + i] = iq < a.ntok ? pm[i] : 0.f;
            }
            if (ok && iq < a.ntok && keep && !(a.dbg & 2)) {
                uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(a.pm) + (((size_t)b * nq + (iq - 1)) * J + j) * NH + 4 * kh);
                *dst = make_uint4(pack2_f16(s4.x, pm[0]), pack2_f16(s4.y, pm[1]), pack2_f16(s4.z, pm[2]), pack2_f16(s4.w, pm[3]));
            }
        }
        __syncthreads();                                                          // P is dead: its region now holds the K tiles
    } else if (a.packed) {
        const float* wv = wsh;
        // entries the key side never reads are not written: slots of planes that lie before the grid for this row (ta < ta0 or tb < tb0, as in
        // rowm_planes: 78 % of the slots at dilation 4) and taps whose key column would be negative.  The key side walks the attending query
        // rows of every key that EXISTS, so it only ever asks for the entries kept here.
        const int ta0 = max(0, a.kf - 1 - f / a.df), tb0 = max(0, a.kh - 1 - y / a.dh);
        const float rkw = 1.f / (float)a.kw, rkh = 1.f / (float)a.kh;
        for (int item = t; item < W * J && !(a.dbg & 512); item += blockDim.x) {
            const int wq = (int)(((float)item + 0.5f) * rJ), j = item - wq * J;
            const int iq = 1 + ry * W + wq;
            bool keep = false;                                                    // (slot 0, <bos>: its dk / dv come from this kernel's own partials, never from the workspace)
            if (j > 0) {
                const int pl = (int)(((float)(j - 1) + 0.5f) * rkw), tc = j - 1 - pl * a.kw;
                const int ta = (int)(((float)pl + 0.5f) * rkh), tb = pl - ta * a.kh;
                keep = ta >= ta0 && tb >= tb0 && wq - (a.kw - 1 - tc) * a.dw >= 0;
            }
            const int ib = item * NH + wq * S3M_PAD;
            const float4 p0 = *reinterpret_cast<const float4*>(SP + ib), p1 = *reinterpret_cast<const float4*>(SP + ib + 4);
            const float4 s0 = *reinterpret_cast<const float4*>(DP + ib), s1 = *reinterpret_cast<const float4*>(DP + ib + 4);
            const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            const float dsv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            uint32_t pw[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 w0 = *reinterpret_cast<const float4*>(wv + g * NH), w1 = *reinterpret_cast<const float4*>(wv + g * NH + 4);
                const float wg[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                float sm = 0.f;
#pragma unroll
                for (int hh = 0; hh < 8; ++hh) sm += wg[hh] * pv[hh];
                if (j == 0) PM0[wq * NH + g] = iq < a.ntok ? sm : 0.f;
                pw[g] = pack2_t<G16>(dsv[g], sm);                                 // low half: ds[head g], high half: P'[head g]  (G16: fp16 halves; ds was clamped in the ds pass)
            }
            if (iq < a.ntok && keep && !gst && !(a.dbg & 2)) {
                uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(a.pm) + (((size_t)b * nq + (iq - 1)) * J + j) * NH);
                dst[0] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
                dst[1] = make_uint4(pw[4], pw[5], pw[6], pw[7]);
            }
        }
        __syncthreads();                                                          // P is dead: its region now holds the K tiles
    }
    {
        const int h = r.wave;
        f32x4 O[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (!(a.dbg & 4)) mfma_band_apply_fast<G16>(a, r, a.k, a.ld, h, DP, smem + r.wave * 4096, O);
        if (r.qok) {
            bf16_t* orow = a.dq + (r.tok0 + r.iq) * a.ldd + h * DH + 4 * r.g4;
            float amax = 0.f;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                *reinterpret_cast<uint2*>(orow + db * 16) = G16 ?
                    make_uint2(pack2_f16_sat_n(O[db][0] * a.scale, O[db][1] * a.scale, amax), pack2_f16_sat_n(O[db][2] * a.scale, O[db][3] * a.scale, amax)) :
                    make_uint2(pack2_rne(O[db][0] * a.scale, O[db][1] * a.scale), pack2_rne(O[db][2] * a.scale, O[db][3] * a.scale));
            if constexpr (G16) f16_sat_commit(amax);
        }
    }
    // <bos> partials of this row: dk0[e] = scale * sum_w ds[w][0][h] q[w][e],  dv0[e] = sum_w P'[w][0][g] dO[w][e]
    // Round 6: 16-byte row pieces.  Thread (wave, eg) takes channels 8 eg .. 8 eg + 7 of query rows 2 wave and 2 wave + 1 (four 16-byte loads;
    // the form before it had one thread per channel issue thirty-two 2-byte loads: 256 vector-memory instructions per workgroup at its tail,
    // 163 us of the 1410-us kernel at dilation 2 in the phase probe, profiles/r06o_s3q_probe.txt), the eight waves' partial sums meet in the
    // LDS region the dq sweep's K tiles have left (fixed order: waves ascending) and thread e writes channel e.
    __syncthreads();                                                              // every wave is done with its K tile
    if (!(a.dbg & 1024)) {
        // records of 16 floats [8 waves][64 eg] = 32 KiB <= r1 (the ds table behind it is still being read); the four 16-byte pieces of a record
        // sit rotated by eg / 4 so that the 16 lanes of a store phase (64-byte pitch) hit 16 different bank quads
        float* part = reinterpret_cast<float*>(smem);
        const int eg = t & 63, wv = t >> 6, h = eg >> 3;
        float sk[8], sv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sk[k] = 0.f; sv[k] = 0.f; }
        uint4 qv[2], dv4[2];
        float cs[2], cp[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int w = 2 * wv + rr, i = 1 + ry * W + w;
            const bool in = i < a.ntok;
            const size_t ic = r.tok0 + (in ? i : a.ntok - 1);
            qv[rr] = *reinterpret_cast<const uint4*>(a.q + ic * a.ld + 8 * eg);
            dv4[rr] = *reinterpret_cast<const uint4*>(a.dO + ic * a.lddo + 8 * eg);
            cs[rr] = in ? DP[w * TS + h] : 0.f;
            cp[rr] = in ? PM0[w * NH + h] : 0.f;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const uint32_t qw[4] = {qv[rr].x, qv[rr].y, qv[rr].z, qv[rr].w}, dw[4] = {dv4[rr].x, dv4[rr].y, dv4[rr].z, dv4[rr].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sk[2 * k] = fmaf(cs[rr], lo_t<G16>(qw[k]), sk[2 * k]); sk[2 * k + 1] = fmaf(cs[rr], hi_t<G16>(qw[k]), sk[2 * k + 1]);
                sv[2 * k] = fmaf(cp[rr], lo_t<G16>(dw[k]), sv[2 * k]); sv[2 * k + 1] = fmaf(cp[rr], hi_t<G16>(dw[k]), sv[2 * k + 1]);
            }
        }
        float4* dst = reinterpret_cast<float4*>(part + (wv * 64 + eg) * 16);
        const int rot = eg >> 2;
        dst[rot & 3] = make_float4(sk[0], sk[1], sk[2], sk[3]); dst[(1 + rot) & 3] = make_float4(sk[4], sk[5], sk[6], sk[7]);
        dst[(2 + rot) & 3] = make_float4(sv[0], sv[1], sv[2], sv[3]); dst[(3 + rot) & 3] = make_float4(sv[4], sv[5], sv[6], sv[7]);
    }
    __syncthreads();
    if (!(a.dbg & 1024)) {
        const float* part = reinterpret_cast<const float*>(smem);
        for (int e = t; e < inner; e += blockDim.x) {
            const int eg = e >> 3, cmp = e & 7, rot = eg >> 2;
            const int ok_ = 4 * (((cmp >> 2) + rot) & 3) + (cmp & 3), ov_ = 4 * ((2 + (cmp >> 2) + rot) & 3) + (cmp & 3);
            float sk = 0.f, sv = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) {
                sk += part[(wv * 64 + eg) * 16 + ok_];
                sv += part[(wv * 64 + eg) * 16 + ov_];
            }
            pk0[e] = a.scale * sk;
            pv0[e] = sv;
        }
    }
}

// MFMA backward, key side: one workgroup = one KEY row of the grid, wave = head.  Every plane's attending query row sends
//   dK^T[d][key] += Q^T[d][query] . ds[query][key]      dV^T[d][key] += dO^T[d][query] . P'[query][key]
// where the (query, key) coefficient is the tap entry of the band (key = query - (kw-1-tc) dw), read from the fp32 workspace
// the query-side kernel wrote.  The q / dO rows of two planes sit in two wave-private transposed tiles; no atomics, every key
// pulls from the queries that attend to it.
// PACKED: the workspace holds (bf16 ds | bf16 P') words (one 4-byte load per coefficient pair, the MFMA operands assembled with byte
// permutes); else two fp32 arrays (two loads, two round-to-nearest conversions): the same operand bits either way.
template <bool PACKED, bool G16 = false>     // G16: fp16 q / dO rows, (fp16 S ds | fp16 P') workspace words, fp16 outputs (S3Args::gs2)
__global__ __launch_bounds__(512, 2) void s3_bwd_kv_mfma_kernel(S3Args a) {
    constexpr int NH = S3M_NH, DH = S3M_DH, W = S3M_W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int pslot[S3M_PLANES + 1], ptok[S3M_PLANES + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g4 = lane >> 4, h = wave;
    const int rows = a.F * a.H;
    const int bid = xcd_row_id();
    const int b = bid / rows, r2 = bid % rows;
    int f, y;
    s3m_row_order(a, r2, f, y);
    const int ry = f * a.H + y;
    const int J = a.kf * a.kh * a.kw + 1, nq = a.ntok - 1;
    if (ry * W + 1 >= a.ntok) return;
    if (t < 64) {
        // the attending query rows, one lane per (ta, tb) candidate, compacted in (ta, tb) order by a ballot (was a serial loop of thread 0)
        const int ta = t / a.kh, tb = t % a.kh;
        const int fq = f + (a.kf - 1 - ta) * a.df, yq = y + (a.kh - 1 - tb) * a.dh;
        const bool ok = t < a.kf * a.kh && fq < a.F && yq < a.H && (fq * a.H + yq) * W + 1 < a.ntok;
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const int pos = __popcll(m & ((1ull << t) - 1ull));
            pslot[pos] = 1 + (ta * a.kh + tb) * a.kw; ptok[pos] = 1 + (fq * a.H + yq) * W;
        }
        if (t == 0) pslot[S3M_PLANES] = __popcll(m);
    }
    __syncthreads();
    const int nplanes = pslot[S3M_PLANES];
    const size_t tok0 = (size_t)b * a.ntok;
    // tap linking the lane's 4 QUERIES 4*g4 + j to key c (query - key = (kw-1-tc) dw): the same in every plane
    int tsel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = (4 * g4 + j) - c;
        tsel[j] = -1;
#pragma unroll
        for (int tc = 0; tc < S3M_KW; ++tc)
            if (tc < a.kw && d == (a.kw - 1 - tc) * a.dw) tsel[j] = tc;
    }
    char* tq = smem + wave * 8192;
    char* td = tq + 4096;
    const int gc = lane & 7, r8 = lane >> 3;
    const bf16_t* qbase = a.q + tok0 * a.ld + h * DH + gc * 8;
    const bf16_t* dbase = a.dO + tok0 * a.lddo + h * DH + gc * 8;
    int woff[4], troff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = vt_off(r8 + 8 * i, gc);
    {
        const int r0 = 4 * g4 + (c >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 16 + ((c & 3) << 2);
            troff[db] = vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8;
        }
    }
    uint4 sq[4], sd[4];
    float cs[8], cp[8];                                                          // ds / P' coefficients of the 8 (plane, query) slots
    uint32_t cw[8];                                                              // ... or (PACKED) their (bf16 ds | bf16 P') words
    // Round 6 (PACKED form): the plane lists in one register each (plane p in lane p, v_readlane with the loop counter), q / dO rows and
    // workspace words addressed as a wave-uniform base + a 32-bit byte offset = a scalar term of the plane + a per-lane constant, planes
    // whose 16 query rows all exist fetched without a mask (see mfma_band_scores_fast; the form before it spent a 64-bit multiply-add chain
    // per row piece and per coefficient word: 16 of them per pair of planes and lane).
    const int vps = lane < nplanes ? pslot[lane] : 0, vpt = lane < nplanes ? ptok[lane] : 0;
    const int hu = __builtin_amdgcn_readfirstlane(h);
    const char* sbq = reinterpret_cast<const char*>(a.q + tok0 * a.ld + hu * DH);
    const char* sbd = reinterpret_cast<const char*>(a.dO + tok0 * a.lddo + hu * DH);
    const char* sbw = reinterpret_cast<const char*>(reinterpret_cast<const uint32_t*>(a.pm) + (size_t)b * nq * J * NH + hu);
    const unsigned ldqb = (unsigned)a.ld * 2u, lddb = (unsigned)a.lddo * 2u;
    const unsigned vq0 = (unsigned)r8 * ldqb + (unsigned)gc * 16u, vq1 = vq0 + 8u * ldqb;
    const unsigned vd0 = (unsigned)r8 * lddb + (unsigned)gc * 16u, vd1 = vd0 + 8u * lddb;
    unsigned cj[4];                                                              // ((4 g4 + j) J + tap) words of 32 bytes: the lane's part of a coefficient's offset
#pragma unroll
    for (int j = 0; j < 4; ++j) cj[j] = 32u * (unsigned)((4 * g4 + j) * J + (tsel[j] < 0 ? 0 : tsel[j]));
    auto fetch = [&](int pi) {
        if constexpr (PACKED) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int pj = pi + kb;
                if (pj >= nplanes) {                                              // (uniform: the odd plane of the last pair)
                    sq[2 * kb] = sq[2 * kb + 1] = sd[2 * kb] = sd[2 * kb + 1] = make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) cw[kb * 4 + j] = 0u;
                    continue;
                }
                const int base = __builtin_amdgcn_readlane(vpt, pj < 64 ? pj : 63);
                const unsigned soq = (unsigned)base * ldqb, sod = (unsigned)base * lddb;
                const unsigned sw = 32u * (unsigned)((base - 1) * J + __builtin_amdgcn_readlane(vps, pj < 64 ? pj : 63));
                const bool nof = (a.dbg & 32) != 0, nog = (a.dbg & 16) != 0;
                if (base + 16 <= a.ntok && !nof && !nog) {
                    sq[2 * kb] = ldg_u4(sbq, soq + vq0); sq[2 * kb + 1] = ldg_u4(sbq, soq + vq1);
                    sd[2 * kb] = ldg_u4(sbd, sod + vd0); sd[2 * kb + 1] = ldg_u4(sbd, sod + vd1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) cw[kb * 4 + j] = tsel[j] >= 0 ? *reinterpret_cast<const uint32_t*>(sbw + (sw + cj[j])) : 0u;
                } else {
                    const bool ok0 = base + r8 < a.ntok && !nof, ok1 = base + r8 + 8 < a.ntok && !nof;
                    sq[2 * kb] = ok0 ? ldg_u4(sbq, soq + vq0) : make_uint4(0, 0, 0, 0); sq[2 * kb + 1] = ok1 ? ldg_u4(sbq, soq + vq1) : make_uint4(0, 0, 0, 0);
                    sd[2 * kb] = ok0 ? ldg_u4(sbd, sod + vd0) : make_uint4(0, 0, 0, 0); sd[2 * kb + 1] = ok1 ? ldg_u4(sbd, sod + vd1) : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        cw[kb * 4 + j] = (tsel[j] >= 0 && base + 4 * g4 + j < a.ntok && !nog) ? *reinterpret_cast<const uint32_t*>(sbw + (sw + cj[j])) : 0u;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pj = pi + (i >> 1);
            const int tok = pj < nplanes ? ptok[pj] + r8 + 8 * (i & 1) : a.ntok;
            const bool ok = tok < a.ntok;
            sq[i] = (ok && !(a.dbg & 32)) ? *reinterpret_cast<const uint4*>(qbase + (size_t)tok * a.ld) : make_uint4(0, 0, 0, 0);
            sd[i] = (ok && !(a.dbg & 32)) ? *reinterpret_cast<const uint4*>(dbase + (size_t)tok * a.lddo) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int pj = pi + kb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float vs = 0.f, vp = 0.f;
                uint32_t vw = 0u;
                if (pj < nplanes && tsel[j] >= 0 && !(a.dbg & 16)) {
                    const int tok = ptok[pj] + 4 * g4 + j;
                    if (tok < a.ntok) {
                        const size_t ci = (((size_t)b * nq + (tok - 1)) * J + pslot[pj] + tsel[j]) * NH + h;
                        if constexpr (PACKED) vw = reinterpret_cast<const uint32_t*>(a.pm)[ci];
                        else { vs = a.ds[ci]; vp = a.pm[ci]; }
                    }
                }
                cs[kb * 4 + j] = vs; cp[kb * 4 + j] = vp; cw[kb * 4 + j] = vw;
            }
        }
    };
    f32x4 dK[4], dV[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dK[db] = dV[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nplanes > 0) fetch(0);
    for (int pi = 0; pi < nplanes; pi += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(tq + woff[i]) = sq[i];
            *reinterpret_cast<uint4*>(td + woff[i]) = sd[i];
        }
        bf16x8 bs, bp;
        if constexpr (PACKED) {
            // low halves -> the ds operand, high halves -> the P' operand (v_perm_b32 selectors: bytes 1,0 of each word / bytes 3,2)
            bs = __builtin_bit_cast(bf16x8, make_uint4(__builtin_amdgcn_perm(cw[1], cw[0], 0x05040100u), __builtin_amdgcn_perm(cw[3], cw[2], 0x05040100u),
                                                       __builtin_amdgcn_perm(cw[5], cw[4], 0x05040100u), __builtin_amdgcn_perm(cw[7], cw[6], 0x05040100u)));
            bp = __builtin_bit_cast(bf16x8, make_uint4(__builtin_amdgcn_perm(cw[1], cw[0], 0x07060302u), __builtin_amdgcn_perm(cw[3], cw[2], 0x07060302u),
                                                       __builtin_amdgcn_perm(cw[5], cw[4], 0x07060302u), __builtin_amdgcn_perm(cw[7], cw[6], 0x07060302u)));
        } else {
            bs = __builtin_bit_cast(bf16x8, make_uint4(pack2_rne(cs[0], cs[1]), pack2_rne(cs[2], cs[3]), pack2_rne(cs[4], cs[5]), pack2_rne(cs[6], cs[7])));
            bp = __builtin_bit_cast(bf16x8, make_uint4(pack2_rne(cp[0], cp[1]), pack2_rne(cp[2], cp[3]), pack2_rne(cp[4], cp[5]), pack2_rne(cp[6], cp[7])));
        }
        if (pi + 2 < nplanes) fetch(pi + 2);                                      // next chunk in flight during the MFMAs
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4 ql = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tq + troff[db]));
            const s16x4 qh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tq + troff[db] + 2048));
            const s16x4 dl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(td + troff[db]));
            const s16x4 dh_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(td + troff[db] + 2048));
            const s16x8 q8 = {ql[0], ql[1], ql[2], ql[3], qh[0], qh[1], qh[2], qh[3]};
            const s16x8 d8 = {dl[0], dl[1], dl[2], dl[3], dh_[0], dh_[1], dh_[2], dh_[3]};
            dK[db] = mfma16<G16>(__builtin_bit_cast(bf16x8, q8), bs, dK[db]);
            dV[db] = mfma16<G16>(__builtin_bit_cast(bf16x8, d8), bp, dV[db]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const int ik = 1 + ry * W + c;
    if (ik < a.ntok) {
        bf16_t* kr = a.dk + (tok0 + ik) * a.ldd + h * DH + 4 * g4;
        bf16_t* vr = a.dv + (tok0 + ik) * a.ldd + h * DH + 4 * g4;
        float amax = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            if constexpr (G16) {
                *reinterpret_cast<uint2*>(kr + db * 16) = make_uint2(pack2_f16_sat_n(dK[db][0] * a.scale, dK[db][1] * a.scale, amax),
                                                                     pack2_f16_sat_n(dK[db][2] * a.scale, dK[db][3] * a.scale, amax));
                *reinterpret_cast<uint2*>(vr + db * 16) = make_uint2(pack2_f16_sat_n(dV[db][0], dV[db][1], amax), pack2_f16_sat_n(dV[db][2], dV[db][3], amax));
            } else {
                *reinterpret_cast<uint2*>(kr + db * 16) = make_uint2(pack2_rne(dK[db][0] * a.scale, dK[db][1] * a.scale),
                                                                     pack2_rne(dK[db][2] * a.scale, dK[db][3] * a.scale));
                *reinterpret_cast<uint2*>(vr + db * 16) = make_uint2(pack2_rne(dV[db][0], dV[db][1]), pack2_rne(dV[db][2], dV[db][3]));
            }
        }
        if constexpr (G16) f16_sat_commit(amax);
    }
}

// MFMA backward, key side, RECOMPUTING form (no ds / P' workspace).  Same mapping as s3_bwd_kv_mfma_kernel (one workgroup = one key
// row, wave = head), one plane (= one attending query row) per iteration:
//   S[q][key]   = Q[h] K[h]^T, dP'[q][key] = dO[h] V[h]^T        2 + 2 MFMAs: A = the plane's q / dO rows out of the wave's tile,
//                                                                 B = this row's k / v fragments, loaded once per workgroup
//   P[h]        = exp(scale S + bias - m[q][h]) / l[q][h]         on the band entries (key = query - tap offset), from the statistics
//                                                                 the query side wrote: 3 floats per (query, head)
//   heads meet through LDS (bf16 x 4 per lane and head, double-buffered by plane parity: ONE workgroup barrier per plane):
//   P'[g]  = sum_h W[g][h] P[h]          (this wave's output head g = its own index)
//   dP[h]  = sum_g W[g][h] dP'[g],       ds[h] = P[h] (dP[h] - delta[q][h])
//   dK^T += Q^T ds,  dV^T += dO^T P'     as before (transposing tile reads; the upper 16 k-slots of the MFMA stay zero)
// What disappears: the 2 x [B][n][J][8] fp32 workspace (3.9 GB written and read back per call at b = 128) -- the backward's HBM
// traffic drops from 2.85x to ~1.5x the algorithmic bytes.  What it costs: one workgroup barrier per plane and ~250 VALU instructions
// per plane and wave; at 180 registers only one workgroup fits a CU.  Measured 15-22 % SLOWER than the workspace form (whole backward,
// tools/attn_bench.py), so it is an option (tuning key 4 = 4), not the default.  The coefficients meet as bf16 (they are packed to bf16 for the MFMAs
// anyway); P of the own head stays fp32 in ds.  LDS: 8 x 4 KiB tiles + 16 KiB exchange = 48 KiB.
__global__ __launch_bounds__(512, 2) void s3_bwd_kv_rc_mfma_kernel(S3Args a) {
    constexpr int NH = S3M_NH, DH = S3M_DH, W = S3M_W;
    constexpr bool G16 = false;                                                  // (the bf16 form only: shares its tail with s3_bwd_kv_mfma_kernel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int pslot[S3M_PLANES + 1], ptok[S3M_PLANES + 1];
    __shared__ float wsh[64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 15, g4 = lane >> 4, h = wave;
    const int rows = a.F * a.H;
    const int bid = xcd_row_id();
    const int b = bid / rows, r2 = bid % rows;
    int f, y;
    s3m_row_order(a, r2, f, y);
    const int ry = f * a.H + y;
    const int nq = a.ntok - 1;
    if (ry * W + 1 >= a.ntok) return;
    if (t < NH * NH) wsh[t] = a.wth[t];
    if (t == 0) {
        int n = 0;
        for (int ta = 0; ta < a.kf; ++ta)
            for (int tb = 0; tb < a.kh; ++tb) {
                const int fq = f + (a.kf - 1 - ta) * a.df, yq = y + (a.kh - 1 - tb) * a.dh;
                if (fq < a.F && yq < a.H && (fq * a.H + yq) * W + 1 < a.ntok) {
                    pslot[n] = 1 + (ta * a.kh + tb) * a.kw; ptok[n] = 1 + (fq * a.H + yq) * W; ++n;
                }
            }
        pslot[S3M_PLANES] = n;
    }
    __syncthreads();
    const int nplanes = pslot[S3M_PLANES];
    const size_t tok0 = (size_t)b * a.ntok;
    // tap linking the lane's 4 QUERIES 4*g4 + j to key c (query - key = (kw-1-tc) dw): the same in every plane
    int tsel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = (4 * g4 + j) - c;
        tsel[j] = -1;
#pragma unroll
        for (int tc = 0; tc < S3M_KW; ++tc)
            if (tc < a.kw && d == (a.kw - 1 - tc) * a.dw) tsel[j] = tc;
    }
    // this key row's k / v as B operands (lane: column = key c, k-group g4 -> channels ks * 32 + g4 * 8 ..)
    const int ik = 1 + ry * W + c;
    const bool kok = ik < a.ntok;
    const bf16_t* krow = a.k + (tok0 + (kok ? ik : 0)) * a.ld + h * DH + g4 * 8;
    const bf16_t* vrow = a.v + (tok0 + (kok ? ik : 0)) * a.ld + h * DH + g4 * 8;
    const bf16x8 kf0 = ldg8(krow, kok), kf1 = ldg8(krow + 32, kok), vf0 = ldg8(vrow, kok), vf1 = ldg8(vrow + 32, kok);
    float wg[8], wt[8];                       // row h of W (P'[h] = sum_hh W[h][hh] P[hh]) and column h (dP[h] = sum_g W[g][h] dP'[g])
#pragma unroll
    for (int e = 0; e < 8; ++e) {             // wave-uniform: kept in scalar registers
        wg[e] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wsh[h * NH + e])));
        wt[e] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wsh[e * NH + h])));
    }
    char* tq = smem + wave * 4096;            // [16 queries][64] bf16 (vt_off swizzle)
    char* td = tq + 2048;
    uint2* XP = reinterpret_cast<uint2*>(smem + 8 * 4096);        // [2][NH][64] P as 4 x bf16
    uint2* XD = XP + 2 * NH * 64;                                  // [2][NH][64] dP'
    const int gc = lane & 7, r8 = lane >> 3;
    const bf16_t* qbase = a.q + tok0 * a.ld + h * DH + gc * 8;
    const bf16_t* dbase = a.dO + tok0 * a.lddo + h * DH + gc * 8;
    const int woff0 = vt_off(r8, gc), woff1 = vt_off(r8 + 8, gc);
    const int aoff0 = vt_off(c, g4), aoff1 = aoff0 ^ 64;                          // A operand: row = query c (lane & 15), chunk ks * 4 + g4 (chunk + 4 = byte ^ 64)
    int troff[4];
    {
        const int r0 = 4 * g4 + (c >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 16 + ((c & 3) << 2);
            troff[db] = vt_off(r0, col >> 3) + ((col >> 2) & 1) * 8;
        }
    }
    const float* stb = a.stats + (size_t)b * nq * NH * 4 + h * 4;
    uint4 sq[2], sd[2];
    float2 st[4];                                                                // (row max, 1 / row sum) of the lane's 4 queries
    float dz[4];                                                                 // their delta
    auto fetch_rows = [&](int p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int tok = ptok[p] + r8 + 8 * i;
            const bool ok = tok < a.ntok;
            const size_t tc = ok ? tok : 0;
            const uint4 vq = *reinterpret_cast<const uint4*>(qbase + tc * a.ld), vd = *reinterpret_cast<const uint4*>(dbase + tc * a.lddo);
            sq[i] = ok ? vq : make_uint4(0, 0, 0, 0);
            sd[i] = ok ? vd : make_uint4(0, 0, 0, 0);
        }
    };
    auto fetch_stats = [&](int p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tok = ptok[p] + 4 * g4 + j;
            const bool ok = tok < a.ntok && tsel[j] >= 0;
            const float* sp = stb + (size_t)((ok ? tok : 1) - 1) * NH * 4;
            const float2 v = *reinterpret_cast<const float2*>(sp);
            const float z = sp[2];
            st[j] = ok ? v : make_float2(0.f, 0.f);                              // inv = 0 -> P = 0 for slots that do not exist
            dz[j] = ok ? z : 0.f;
        }
    };
    f32x4 dK[4], dV[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dK[db] = dV[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nplanes > 0) { fetch_rows(0); fetch_stats(0); }
    for (int p = 0; p < nplanes; ++p) {
        *reinterpret_cast<uint4*>(tq + woff0) = sq[0];
        *reinterpret_cast<uint4*>(tq + woff1) = sq[1];
        *reinterpret_cast<uint4*>(td + woff0) = sd[0];
        *reinterpret_cast<uint4*>(td + woff1) = sd[1];
        const int jb = pslot[p];
        if (p + 1 < nplanes) fetch_rows(p + 1);                                   // the next plane's rows are in flight below
        __builtin_amdgcn_wave_barrier();                                          // LDS is in-order per wave: the tiles are complete
        const bf16x8 qa0 = *reinterpret_cast<const bf16x8*>(tq + aoff0), qa1 = *reinterpret_cast<const bf16x8*>(tq + aoff1);
        const bf16x8 da0 = *reinterpret_cast<const bf16x8*>(td + aoff0), da1 = *reinterpret_cast<const bf16x8*>(td + aoff1);
        f32x4 S = {0.f, 0.f, 0.f, 0.f}, Dp = {0.f, 0.f, 0.f, 0.f};
        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, kf0, S, 0, 0, 0);        // S[query 4 g4 + r][key c]
        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, kf1, S, 0, 0, 0);
        Dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, vf0, Dp, 0, 0, 0);      // dP'[h][query][key] = dO[h] . V[h]
        Dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, vf1, Dp, 0, 0, 0);
        float P[4], dpp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = tsel[j] >= 0 && kok;
            const int ts = tsel[j] < 0 ? 0 : tsel[j];
            const float bv = a.bias ? a.bias[(jb + ts) * NH + h] : 0.f;
            const float e = __expf(S[j] * a.scale + bv - st[j].x) * st[j].y;        // (inv = 0 where the query / slot does not exist)
            P[j] = on ? e : 0.f;
            dpp[j] = on ? Dp[j] : 0.f;
        }
        float dzc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dzc[j] = dz[j];
        if (p + 1 < nplanes) fetch_stats(p + 1);                                  // (the statistics of this plane are consumed)
        const int par = p & 1;
        XP[(par * NH + h) * 64 + lane] = make_uint2(pack2_rne(P[0], P[1]), pack2_rne(P[2], P[3]));
        XD[(par * NH + h) * 64 + lane] = make_uint2(pack2_rne(dpp[0], dpp[1]), pack2_rne(dpp[2], dpp[3]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float pm[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const uint2 xp = XP[(par * NH + e) * 64 + lane], xd = XD[(par * NH + e) * 64 + lane];
            pm[0] += wg[e] * lo_f(xp.x); pm[1] += wg[e] * hi_f(xp.x); pm[2] += wg[e] * lo_f(xp.y); pm[3] += wg[e] * hi_f(xp.y);
            dp[0] += wt[e] * lo_f(xd.x); dp[1] += wt[e] * hi_f(xd.x); dp[2] += wt[e] * lo_f(xd.y); dp[3] += wt[e] * hi_f(xd.y);
        }
        float dsv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dsv[j] = P[j] * (dp[j] - dzc[j]);
        const bf16x8 bs = __builtin_bit_cast(bf16x8, make_uint4(pack2_rne(dsv[0], dsv[1]), pack2_rne(dsv[2], dsv[3]), 0u, 0u));
        const bf16x8 bp = __builtin_bit_cast(bf16x8, make_uint4(pack2_rne(pm[0], pm[1]), pack2_rne(pm[2], pm[3]), 0u, 0u));
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4 ql = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tq + troff[db]));
            const s16x4 dl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(td + troff[db]));
            const s16x8 q8 = {ql[0], ql[1], ql[2], ql[3], 0, 0, 0, 0};
            const s16x8 d8 = {dl[0], dl[1], dl[2], dl[3], 0, 0, 0, 0};
            dK[db] = mfma16<G16>(__builtin_bit_cast(bf16x8, q8), bs, dK[db]);
            dV[db] = mfma16<G16>(__builtin_bit_cast(bf16x8, d8), bp, dV[db]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (kok) {
        bf16_t* kr = a.dk + (tok0 + ik) * a.ldd + h * DH + 4 * g4;
        bf16_t* vr = a.dv + (tok0 + ik) * a.ldd + h * DH + 4 * g4;
        float amax = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            if constexpr (G16) {
                *reinterpret_cast<uint2*>(kr + db * 16) = make_uint2(pack2_f16_sat_n(dK[db][0] * a.scale, dK[db][1] * a.scale, amax),
                                                                     pack2_f16_sat_n(dK[db][2] * a.scale, dK[db][3] * a.scale, amax));
                *reinterpret_cast<uint2*>(vr + db * 16) = make_uint2(pack2_f16_sat_n(dV[db][0], dV[db][1], amax), pack2_f16_sat_n(dV[db][2], dV[db][3], amax));
            } else {
                *reinterpret_cast<uint2*>(kr + db * 16) = make_uint2(pack2_rne(dK[db][0] * a.scale, dK[db][1] * a.scale),
                                                                     pack2_rne(dK[db][2] * a.scale, dK[db][3] * a.scale));
                *reinterpret_cast<uint2*>(vr + db * 16) = make_uint2(pack2_rne(dV[db][0], dV[db][1]), pack2_rne(dV[db][2], dV[db][3]));
            }
        }
        if constexpr (G16) f16_sat_commit(amax);
    }
}

void fill_geom(S3Args& a, const amdnuwa_s3_geom* g) {
    a.B = g->B; a.ntok = g->ntok; a.F = g->F; a.H = g->H; a.W = g->W; a.kf = g->kf; a.kh = g->kh; a.kw = g->kw;
    a.df = g->df; a.dh = g->dh; a.dw = g->dw; a.NH = g->heads; a.scale = g->scale; a.bias = g->rel_bias;
    a.ymajor = (g_amdnuwa_tuning[3] & 2) ? 0 : 1;
    // causal: every tap at or before the query (np.py:427); symmetric 'same' window otherwise (np.py:429)
    a.of = g->noncausal ? (g->kf - 1) / 2 : g->kf - 1;
    a.oh = g->noncausal ? (g->kh - 1) / 2 : g->kh - 1;
    a.ow = g->noncausal ? (g->kw - 1) / 2 : g->kw - 1;
    a.xmode = 0; a.FK = g->F; a.kvrows = g->ntok; a.kvoff = 1; a.kmask = nullptr;
}
// self-attention: keys / values are the query sequence, slot 0 = its <bos> row
void self_kv(S3Args& a) {
    a.ldk = a.ld; a.lddk = a.ldd;
    a.k0 = a.k; a.k0l = a.kl; a.v0 = a.v; a.v0l = a.vl; a.k0_bs = (long long)a.ntok * a.ld;
}
int block_threads(const amdnuwa_s3_geom* g) { return ((g->W * g->heads * 4 + 63) / 64) * 64; }

// query rows per workgroup of the MFMA kernels (tuning key 16: 0 = auto, 1 / 2 / 4 = forced where the geometry allows it): a tile is
// ROWS rows of one residue class of y modulo the dilation, so H must split into dh * ROWS
int s3_tile_rows(const amdnuwa_s3_geom* g) {
    int want = g_amdnuwa_tuning[16];
    if (want == 0) want = S3T_AUTO_ROWS;
    for (int rws = want >= 4 ? 4 : (want >= 2 ? 2 : 1); rws > 1; rws >>= 1)
        if (g->dh > 0 && g->H % (g->dh * rws) == 0 && g->kf * (g->kh + rws - 1) <= S3T_MAXK) return rws;
    return 1;
}
template <bool F16>
int s3_fwd_tile_launch(const S3Args& a, const amdnuwa_s3_geom* g, int tr, hipStream_t stream) {
    const int J = g->kf * g->kh * g->kw + 1;
    const size_t lm = (size_t)tr * 16 * (J * 8 + 4) * sizeof(float) + 8 * 4096;
    const dim3 grid(g->B * g->F * (g->H / tr));
#define S3T_LAUNCH(R_, B_)                                                                                        \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)s3_fwd_tile_kernel<R_, F16, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm); \
        hipLaunchKernelGGL((s3_fwd_tile_kernel<R_, F16, B_>), grid, dim3(512), lm, stream, a);                    \
    } while (0)
    if (tr == 4) { if (a.bias) S3T_LAUNCH(4, true); else S3T_LAUNCH(4, false); }
    else { if (a.bias) S3T_LAUNCH(2, true); else S3T_LAUNCH(2, false); }
#undef S3T_LAUNCH
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

}  // namespace

extern "C" int amdnuwa_sparse3dna_fwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                      const uint16_t* q_lo, const uint16_t* k_lo, const uint16_t* v_lo, int ld,
                                      const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, hipStream_t stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    if (s3_lds_need(g, k_lo != nullptr) > S3_LDS_MAX) return AMDNUWA_ERR_UNSUPPORTED;
    if (!q || !k || !v || !w_th || !o || ld % 8 || ldo % 8) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    S3Args a{};
    fill_geom(a, g);
    a.q = q; a.k = k; a.v = v; a.ql = q_lo; a.kl = k_lo; a.vl = v_lo; a.ld = ld;
    a.o = o; a.ol = o_lo; a.ldo = ldo; a.wth = w_th;
    self_kv(a);
    const int J = g->kf * g->kh * g->kw + 1;
    if ((k_lo != nullptr) && (!q_lo || !v_lo)) return AMDNUWA_ERR_ARG;
    // tuning key 3: 0 = one key row per staging round (measured faster: 4 resident workgroups per CU),
    //               1 = stage the kh rows of a tap frame at once (needs kh <= KHMAX)
    const bool slab = false;   // (slab staging of a whole tap frame measured slower than row staging: retired)
    const size_t lds = (size_t)g->W * g->heads * g->dim_head * (k_lo ? 4 : 2) * (slab ? g->kh : 1) + (size_t)g->W * J * g->heads * 4;
    dim3 grid(g->B * g->F * g->H), block(block_threads(g));
#define S3F(DH_, LO_)                                                                                             \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)s3_fwd_kernel<DH_, LO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((s3_fwd_kernel<DH_, LO_>), grid, block, lds, stream, a);                               \
    } while (0)
    const bool lo_mode = k_lo != nullptr;
    // MFMA forward (tuning key 3: 1 = keep the dot2 kernel): bf16 operands, 16 queries per grid row, 8 heads x 64
    if (!lo_mode && !g->noncausal && !(g_amdnuwa_tuning[3] & 1) && g->W == 16 && g->heads == 8 && g->dim_head == 64 && g->kw <= S3M_KW &&
        g->kf * g->kh <= S3M_PLANES && ld % 8 == 0 && ldo % 4 == 0) {
        a.dbg = g_amdnuwa_tuning[9];
        const int tr = s3_tile_rows(g);
        if (tr > 1) return s3_fwd_tile_launch<false>(a, g, tr, stream);
        const size_t lm = (size_t)16 * (J * 8 + 4) * sizeof(float) + 8 * 4096;
        (void)hipFuncSetAttribute((const void*)s3_fwd_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
        hipLaunchKernelGGL(s3_fwd_mfma_kernel<false>, grid, dim3(512), lm, stream, a);
        LAUNCH_CHECK();
        return AMDNUWA_OK;
    }
    if (g->dim_head == 64) { if (lo_mode) S3F(64, true); else S3F(64, false); }
    else { if (lo_mode) S3F(32, true); else S3F(32, false); }
#undef S3F
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

// the geometry of the MFMA band kernels: the causal decoder window on a 16-wide grid, 8 heads x 64
static bool s3_mfma_geom(const amdnuwa_s3_geom* g) {
    return !g->noncausal && g->W == 16 && g->heads == 8 && g->dim_head == 64 && g->kw <= S3M_KW && g->kf * g->kh <= S3M_PLANES;
}
extern "C" int amdnuwa_s3_f16_supported(const amdnuwa_s3_geom* g) { return check_geom(g) == AMDNUWA_OK && s3_mfma_geom(g) ? 1 : 0; }

extern "C" int amdnuwa_sparse3dna_fwd_f16(const amdnuwa_s3_geom* g, const uint16_t* q_f16, const uint16_t* k_f16, const uint16_t* v_f16,
                                          int ld, const float* w_th, uint16_t* o, uint16_t* o_lo, int ldo, int o_lo_f16, hipStream_t stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    if (!s3_mfma_geom(g)) return AMDNUWA_ERR_UNSUPPORTED;
    if (!q_f16 || !k_f16 || !v_f16 || !w_th || ld % 8 || ldo % 8) return AMDNUWA_ERR_ARG;
    if (!o && !(o_lo && o_lo_f16)) return AMDNUWA_ERR_ARG;        // o == NULL: the fp16 copy is the only output
    if (g->B <= 0) return AMDNUWA_OK;
    S3Args a{};
    fill_geom(a, g);
    a.q = q_f16; a.k = k_f16; a.v = v_f16; a.ld = ld;
    a.o = o; a.ol = o_lo; a.ldo = ldo; a.wth = w_th; a.ol_f16 = (o_lo && o_lo_f16) ? 1 : 0;
    a.dbg = g_amdnuwa_tuning[9];
    self_kv(a);
    const int J = g->kf * g->kh * g->kw + 1;
    {
        const int tr = s3_tile_rows(g);
        if (tr > 1) return s3_fwd_tile_launch<true>(a, g, tr, stream);
    }
    const size_t lm = (size_t)16 * (J * 8 + 4) * sizeof(float) + 8 * 4096;
    (void)hipFuncSetAttribute((const void*)s3_fwd_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
    hipLaunchKernelGGL(s3_fwd_mfma_kernel<true>, dim3(g->B * g->F * g->H), dim3(512), lm, stream, a);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_s3_supported(const amdnuwa_s3_geom* g, int lo_operands) {
    if (check_geom(g)) return 0;
    return s3_lds_need(g, lo_operands != 0) <= S3_LDS_MAX ? 1 : 0;
}

extern "C" size_t amdnuwa_sparse3dna_bwd_workspace_bytes(const amdnuwa_s3_geom* g) {
    if (check_geom(g)) return 0;
    const size_t J = (size_t)g->kf * g->kh * g->kw + 1, nq = g->ntok - 1, rows = (size_t)g->B * g->F * g->H;
    const size_t inner = (size_t)g->heads * g->dim_head;
    return (2 * (size_t)g->B * nq * J * g->heads + rows * g->heads * g->heads + 2 * rows * inner) * sizeof(float) + 256 +
           amdnuwa_colsum_workspace_bytes((long long)g->B * nq, (int)(J * g->heads));
}

extern "C" int amdnuwa_sparse3dna_bwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                      const uint16_t* q_lo, const uint16_t* k_lo, const uint16_t* v_lo, int ld,
                                      const float* w_th, const uint16_t* dO, const uint16_t* dO_lo, int lddo,
                                      uint16_t* dq, uint16_t* dk, uint16_t* dv, uint16_t* dq_lo, uint16_t* dk_lo,
                                      uint16_t* dv_lo, int ldd, float* dw_th, int accumulate, void* workspace,
                                      size_t workspace_bytes, hipStream_t stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    if (s3_lds_need(g, k_lo != nullptr) > S3_LDS_MAX) return AMDNUWA_ERR_UNSUPPORTED;
    if (!q || !k || !v || !w_th || !dO || !dq || !dk || !dv || !dw_th || ld % 8 || lddo % 8 || ldd % 8) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_sparse3dna_bwd_workspace_bytes(g)) return AMDNUWA_ERR_WORKSPACE;
    if (g->B <= 0) return AMDNUWA_OK;
    S3Args a{};
    fill_geom(a, g);
    a.q = q; a.k = k; a.v = v; a.ql = q_lo; a.kl = k_lo; a.vl = v_lo; a.ld = ld; a.wth = w_th;
    a.dO = dO; a.dOl = dO_lo; a.lddo = lddo;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dql = dq_lo; a.dkl = dk_lo; a.dvl = dv_lo; a.ldd = ldd;
    a.dwth = dw_th; a.accumulate = accumulate;
    a.dbg = g_amdnuwa_tuning[17];
    a.sep_passes = g_amdnuwa_tuning[19] == 1;
    self_kv(a);
    const size_t J = (size_t)g->kf * g->kh * g->kw + 1, nq = g->ntok - 1, rows = (size_t)g->B * g->F * g->H;
    const size_t inner = (size_t)g->heads * g->dim_head;
    float* ws = (float*)workspace;
    a.ds = ws; ws += (size_t)g->B * nq * J * g->heads;
    a.pm = ws; ws += (size_t)g->B * nq * J * g->heads;
    a.part_th = ws; ws += rows * g->heads * g->heads;
    a.part_k0 = ws; ws += rows * inner;
    a.part_v0 = ws;
    const size_t nsp = (size_t)g->W * J * g->heads;
    // bwd_q LDS: stage (hi+lo) + SP + DP + RED(8*64); the <bos> partial reduction reuses SP+DP: needs 2*W*inner <= 2*nsp
    size_t spdp = 2 * nsp;
    if (spdp < (size_t)g->W * inner) spdp = (size_t)g->W * inner;
    const bool has_lo = k_lo != nullptr;
    if (has_lo && (!q_lo || !v_lo)) return AMDNUWA_ERR_ARG;
    // tuning key 4: 1 = slab staging in bwd_q too (more LDS -> one workgroup per CU), 0 = one row per round
    const bool slab = false;
    const size_t lds_q = (size_t)g->W * g->heads * g->dim_head * (has_lo ? 4 : 2) * (slab ? g->kh : 1) + (spdp + 8 * 64) * 4;
    const size_t lds_kv = (size_t)g->W * g->heads * g->dim_head * 8;
    dim3 grid((unsigned)rows), block(block_threads(g));
    // MFMA query-side kernel (tuning key 4: 1 = keep the dot2 kernel): bf16 operands, 16 queries per grid row, 8 heads x 64
    const bool q_mfma = !has_lo && !dO_lo && !g->noncausal && g_amdnuwa_tuning[4] != 1 && g->W == 16 && g->heads == 8 && g->dim_head == 64 &&
                        g->kw <= S3M_KW && g->kf * g->kh <= S3M_PLANES && ld % 8 == 0 && lddo % 8 == 0 && ldd % 4 == 0;
    // recomputing key side (tuning key 4 = 4; OFF by default: it removes the 3.9 GB workspace round trip -- backward traffic 2.85x ->
    // 1.5x algorithmic -- but runs 15-22 % slower: its per-plane head exchange serialises the eight waves and its 180 registers allow one
    // workgroup per CU).  Needs the MFMA query side; d(rel_bias) is a column sum of the ds workspace.
    const bool kv_rc = q_mfma && g_amdnuwa_tuning[4] == 4 && !g->d_rel_bias;
    a.stats = kv_rc ? a.ds : nullptr;                                              // (lives where the workspace would: B*nq*NH*4 floats <= B*nq*J*NH)
    // packed (bf16 ds | bf16 P') workspace: MFMA query side + MFMA key side, no d(rel_bias) column sum over ds (tuning key 24 = 1: the fp32 pair)
    a.packed = (q_mfma && g_amdnuwa_tuning[4] != 2 && !g->d_rel_bias && g_amdnuwa_tuning[24] != 1 && !a.sep_passes) ? 1 : 0;
    const size_t nspm = (size_t)g->W * (J * g->heads + 4);                          // the MFMA kernels' padded tables (s3m_ts)
    const size_t lds_qm = (nspm * 4 > 8 * 4096 ? nspm * 4 : 8 * 4096) + nspm * 4 + (8 * 64 + 16 * 8) * 4 + 8 * 2048;   // + the score staging tiles
#define S3B(DH_, LO_)                                                                                             \
    do {                                                                                                          \
        if (q_mfma && a.bias) {                                                                                   \
            (void)hipFuncSetAttribute((const void*)s3_bwd_q_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qm); \
            hipLaunchKernelGGL(s3_bwd_q_mfma_kernel<true>, grid, dim3(512), lds_qm, stream, a);                   \
        } else if (q_mfma) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)s3_bwd_q_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qm); \
            hipLaunchKernelGGL(s3_bwd_q_mfma_kernel<false>, grid, dim3(512), lds_qm, stream, a);                  \
        } else {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)s3_bwd_q_kernel<DH_, LO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q); \
            hipLaunchKernelGGL((s3_bwd_q_kernel<DH_, LO_>), grid, block, lds_q, stream, a);                       \
        }                                                                                                         \
        LAUNCH_CHECK();                                                                                           \
        if (kv_rc) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)s3_bwd_kv_rc_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 + 16384); \
            hipLaunchKernelGGL(s3_bwd_kv_rc_mfma_kernel, grid, dim3(512), 8 * 4096 + 16384, stream, a);           \
        } else if (q_mfma && g_amdnuwa_tuning[4] != 2 && a.packed) {                                              \
            (void)hipFuncSetAttribute((const void*)s3_bwd_kv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192); \
            hipLaunchKernelGGL(s3_bwd_kv_mfma_kernel<true>, grid, dim3(512), 8 * 8192, stream, a);                \
        } else if (q_mfma && g_amdnuwa_tuning[4] != 2) {                                                          \
            (void)hipFuncSetAttribute((const void*)s3_bwd_kv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192); \
            hipLaunchKernelGGL(s3_bwd_kv_mfma_kernel<false>, grid, dim3(512), 8 * 8192, stream, a);               \
        } else {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)s3_bwd_kv_kernel<DH_, LO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv); \
            hipLaunchKernelGGL((s3_bwd_kv_kernel<DH_, LO_>), grid, block, lds_kv, stream, a);                     \
        }                                                                                                         \
    } while (0)
    const bool lo_mode = has_lo || dO_lo != nullptr;
    if (lo_mode && (!has_lo || !dO_lo)) return AMDNUWA_ERR_ARG;     // parity mode needs lo parts for q/k/v AND dO
    if (g->dim_head == 64) { if (lo_mode) S3B(64, true); else S3B(64, false); }
    else { if (lo_mode) S3B(32, true); else S3B(32, false); }
#undef S3B
    LAUNCH_CHECK();
    hipLaunchKernelGGL(s3_bwd_fin_kernel, dim3(g->B * (((int)inner + 63) / 64) + (g->heads * g->heads + 15) / 16), dim3(1024), 0, stream, a, g->dim_head);
    LAUNCH_CHECK();
    if (g->d_rel_bias) {         // d(bias)[j][h] = sum over every (sample, query) of ds[.][j][h]: fixed-order column sums of the ds workspace
        float* cws = a.part_v0 + rows * inner;
        const int rc2 = amdnuwa_colsum(a.ds, g->d_rel_bias, (long long)g->B * nq, (int)(J * g->heads), 0, cws,
                                       amdnuwa_colsum_workspace_bytes((long long)g->B * nq, (int)(J * g->heads)), stream);
        if (rc2) return rc2;
    }
    return AMDNUWA_OK;
}

// fp16-gradient form of the backward ('bf16x3-fwd' with block class 's' of AMDNUWA_BWD_F16): q / k / v are the fp16 copies the forward core
// read (the ONLY 16-bit copies the block keeps), dO = fp16(S dO), dq / dk / dv leave as fp16(S gradient), dW_th as fp32 without the factor.
// scale2 = device {S, 1 / S}.  MFMA band kernels + packed workspace only (the causal decoder window, no relative-position bias):
// AMDNUWA_ERR_UNSUPPORTED otherwise (amdnuwa_sparse3dna_bwd_f16_supported).
extern "C" int amdnuwa_sparse3dna_bwd_f16_supported(const amdnuwa_s3_geom* g) {
    return check_geom(g) == AMDNUWA_OK && s3_mfma_geom(g) && !g->rel_bias && !g->d_rel_bias ? 1 : 0;
}
extern "C" int amdnuwa_sparse3dna_bwd_f16(const amdnuwa_s3_geom* g, const uint16_t* q_f16, const uint16_t* k_f16, const uint16_t* v_f16, int ld,
                                          const float* w_th, const uint16_t* dO_f16, int lddo, uint16_t* dq_f16, uint16_t* dk_f16,
                                          uint16_t* dv_f16, int ldd, float* dw_th, int accumulate, const float* scale2, void* workspace,
                                          size_t workspace_bytes, hipStream_t stream) {
    int rc = check_geom(g);
    if (rc) return rc;
    if (!amdnuwa_sparse3dna_bwd_f16_supported(g)) return AMDNUWA_ERR_UNSUPPORTED;
    if (!q_f16 || !k_f16 || !v_f16 || !w_th || !dO_f16 || !dq_f16 || !dk_f16 || !dv_f16 || !dw_th || !scale2 || ld % 8 || lddo % 8 || ldd % 8)
        return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_sparse3dna_bwd_workspace_bytes(g)) return AMDNUWA_ERR_WORKSPACE;
    if (g->B <= 0) return AMDNUWA_OK;
    S3Args a{};
    fill_geom(a, g);
    a.q = q_f16; a.k = k_f16; a.v = v_f16; a.ld = ld; a.wth = w_th;
    a.dO = dO_f16; a.lddo = lddo;
    a.dq = dq_f16; a.dk = dk_f16; a.dv = dv_f16; a.ldd = ldd;
    a.dwth = dw_th; a.accumulate = accumulate; a.gs2 = scale2;
    a.dbg = g_amdnuwa_tuning[17];
    self_kv(a);
    const size_t J = (size_t)g->kf * g->kh * g->kw + 1, nq = g->ntok - 1, rows = (size_t)g->B * g->F * g->H;
    const size_t inner = (size_t)g->heads * g->dim_head;
    float* ws = (float*)workspace;
    a.ds = ws; ws += (size_t)g->B * nq * J * g->heads;
    a.pm = ws; ws += (size_t)g->B * nq * J * g->heads;
    a.part_th = ws; ws += rows * g->heads * g->heads;
    a.part_k0 = ws; ws += rows * inner;
    a.part_v0 = ws;
    a.packed = 1;
    const size_t nspm = (size_t)g->W * (J * g->heads + 4);
    const size_t lds_qm = (nspm * 4 > 8 * 4096 ? nspm * 4 : 8 * 4096) + nspm * 4 + (8 * 64 + 16 * 8) * 4 + 8 * 2048;
    const dim3 grid((unsigned)rows);
    (void)hipFuncSetAttribute((const void*)s3_bwd_q_mfma_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qm);
    hipLaunchKernelGGL((s3_bwd_q_mfma_kernel<false, true>), grid, dim3(512), lds_qm, stream, a);
    LAUNCH_CHECK();
    (void)hipFuncSetAttribute((const void*)s3_bwd_kv_mfma_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192);
    hipLaunchKernelGGL((s3_bwd_kv_mfma_kernel<true, true>), grid, dim3(512), 8 * 8192, stream, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(s3_bwd_fin_kernel, dim3(g->B * (((int)inner + 63) / 64) + (g->heads * g->heads + 15) / 16), dim3(1024), 0, stream, a, g->dim_head);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

// ------------------------------------------------------------------------------------------------
// SparseCross2DNA (np.py:761-901; NUWASketch's decoder cross-attention): the video queries of grid position (y, x) attend, in
// EVERY sketch frame, to the k x k neighbourhood of (y, x) (+ a learned null key / value), with a key mask, fp32 softmax and talking
// heads.  Same kernels as the symmetric Sparse3DNA with the key / value side pointed at the context (S3Args::xmode).  Query row 0
// (<bos>, which attends to every context token without talking heads) is not touched here: the caller owns o[b*ntok] / dq[b*ntok].
// ------------------------------------------------------------------------------------------------
namespace {
int cross_setup(S3Args& a, const amdnuwa_s3_geom* g, int ctx_rows, const uint16_t* k, const uint16_t* v, const uint16_t* k_lo,
                const uint16_t* v_lo, int ldkv, const uint16_t* null_k, const uint16_t* null_k_lo, const uint16_t* null_v,
                const uint16_t* null_v_lo, const uint8_t* key_mask) {
    int rc = check_geom(g);
    if (rc) return rc;
    if (s3_lds_need(g, k_lo != nullptr) > S3_LDS_MAX) return AMDNUWA_ERR_UNSUPPORTED;
    if (!k || !v || !null_k || !null_v || ldkv % 8) return AMDNUWA_ERR_ARG;
    if (g->rel_bias || ctx_rows != g->kf * g->H * g->W) return AMDNUWA_ERR_ARG;
    if ((k_lo != nullptr) != (v_lo != nullptr) || (k_lo != nullptr) != (null_k_lo != nullptr) || (k_lo != nullptr) != (null_v_lo != nullptr))
        return AMDNUWA_ERR_ARG;
    fill_geom(a, g);
    a.xmode = 1; a.FK = g->kf; a.kvrows = ctx_rows; a.kvoff = 0; a.kmask = key_mask;
    a.of = 0; a.df = 1;                                            // frame tap a = context frame a
    a.oh = (g->kh - 1) / 2; a.ow = (g->kw - 1) / 2;                // 'same' padding on the feature map (np.py:789)
    a.k = k; a.v = v; a.kl = k_lo; a.vl = v_lo; a.ldk = ldkv;
    a.k0 = null_k; a.k0l = null_k_lo; a.v0 = null_v; a.v0l = null_v_lo; a.k0_bs = 0;
    return AMDNUWA_OK;
}
}  // namespace

extern "C" int amdnuwa_cross2dna_fwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq, int ctx_rows,
                                     const uint16_t* k, const uint16_t* v, const uint16_t* k_lo, const uint16_t* v_lo, int ldkv,
                                     const uint16_t* null_k, const uint16_t* null_k_lo, const uint16_t* null_v,
                                     const uint16_t* null_v_lo, const uint8_t* key_mask, const float* w_th, uint16_t* o,
                                     uint16_t* o_lo, int ldo, hipStream_t stream) {
    S3Args a{};
    int rc = cross_setup(a, g, ctx_rows, k, v, k_lo, v_lo, ldkv, null_k, null_k_lo, null_v, null_v_lo, key_mask);
    if (rc) return rc;
    if (!q || !w_th || !o || ldq % 8 || ldo % 8 || (k_lo != nullptr) != (q_lo != nullptr)) return AMDNUWA_ERR_ARG;
    if (g->B <= 0) return AMDNUWA_OK;
    a.q = q; a.ql = q_lo; a.ld = ldq; a.o = o; a.ol = o_lo; a.ldo = ldo; a.wth = w_th;
    const int J = g->kf * g->kh * g->kw + 1;
    const bool lo_mode = k_lo != nullptr;
    const size_t lds = (size_t)g->W * g->heads * g->dim_head * (lo_mode ? 4 : 2) + (size_t)g->W * J * g->heads * 4;
    dim3 grid(g->B * g->F * g->H), block(block_threads(g));
#define S3F(DH_, LO_)                                                                                             \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)s3_fwd_kernel<DH_, LO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((s3_fwd_kernel<DH_, LO_>), grid, block, lds, stream, a);                               \
    } while (0)
    if (g->dim_head == 64) { if (lo_mode) S3F(64, true); else S3F(64, false); }
    else { if (lo_mode) S3F(32, true); else S3F(32, false); }
#undef S3F
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" size_t amdnuwa_cross2dna_bwd_workspace_bytes(const amdnuwa_s3_geom* g) { return amdnuwa_sparse3dna_bwd_workspace_bytes(g); }

extern "C" int amdnuwa_cross2dna_bwd(const amdnuwa_s3_geom* g, const uint16_t* q, const uint16_t* q_lo, int ldq, int ctx_rows,
                                     const uint16_t* k, const uint16_t* v, const uint16_t* k_lo, const uint16_t* v_lo, int ldkv,
                                     const uint16_t* null_k, const uint16_t* null_k_lo, const uint16_t* null_v,
                                     const uint16_t* null_v_lo, const uint8_t* key_mask, const float* w_th, const uint16_t* dO,
                                     const uint16_t* dO_lo, int lddo, uint16_t* dq, uint16_t* dq_lo, int lddq, uint16_t* dk,
                                     uint16_t* dv, uint16_t* dk_lo, uint16_t* dv_lo, int lddkv, float* d_null_k, float* d_null_v,
                                     float* dw_th, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    S3Args a{};
    int rc = cross_setup(a, g, ctx_rows, k, v, k_lo, v_lo, ldkv, null_k, null_k_lo, null_v, null_v_lo, key_mask);
    if (rc) return rc;
    if (!q || !w_th || !dO || !dq || !dk || !dv || !d_null_k || !d_null_v || !dw_th || ldq % 8 || lddo % 8 || lddq % 8 || lddkv % 8)
        return AMDNUWA_ERR_ARG;
    const bool lo_mode = k_lo != nullptr;
    if (lo_mode != (q_lo != nullptr) || lo_mode != (dO_lo != nullptr)) return AMDNUWA_ERR_ARG;
    if (!workspace || workspace_bytes < amdnuwa_cross2dna_bwd_workspace_bytes(g)) return AMDNUWA_ERR_WORKSPACE;
    if (g->B <= 0) return AMDNUWA_OK;
    a.q = q; a.ql = q_lo; a.ld = ldq; a.wth = w_th;
    a.dO = dO; a.dOl = dO_lo; a.lddo = lddo;
    a.dq = dq; a.dql = dq_lo; a.ldd = lddq; a.dk = dk; a.dv = dv; a.dkl = dk_lo; a.dvl = dv_lo; a.lddk = lddkv;
    a.dwth = dw_th; a.accumulate = 0; a.dnull_k = d_null_k; a.dnull_v = d_null_v;
    const size_t J = (size_t)g->kf * g->kh * g->kw + 1, nq = g->ntok - 1, rows = (size_t)g->B * g->F * g->H;
    const size_t inner = (size_t)g->heads * g->dim_head;
    float* ws = (float*)workspace;
    a.ds = ws; ws += (size_t)g->B * nq * J * g->heads;
    a.pm = ws; ws += (size_t)g->B * nq * J * g->heads;
    a.part_th = ws; ws += rows * g->heads * g->heads;
    a.part_k0 = ws; ws += rows * inner;
    a.part_v0 = ws;
    const size_t nsp = (size_t)g->W * J * g->heads;
    size_t spdp = 2 * nsp;
    if (spdp < (size_t)g->W * inner) spdp = (size_t)g->W * inner;
    const size_t lds_q = (size_t)g->W * g->heads * g->dim_head * (lo_mode ? 4 : 2) + (spdp + 8 * 64) * 4;
    const size_t lds_kv = (size_t)g->W * g->heads * g->dim_head * 8;
    dim3 grid_q((unsigned)rows), grid_kv((unsigned)(g->B * g->kf * g->H)), block(block_threads(g));
#define S3XB(DH_, LO_)                                                                                            \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)s3_bwd_q_kernel<DH_, LO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q); \
        hipLaunchKernelGGL((s3_bwd_q_kernel<DH_, LO_>), grid_q, block, lds_q, stream, a);                         \
        LAUNCH_CHECK();                                                                                           \
        (void)hipFuncSetAttribute((const void*)s3_bwd_kv_kernel<DH_, LO_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv); \
        hipLaunchKernelGGL((s3_bwd_kv_kernel<DH_, LO_>), grid_kv, block, lds_kv, stream, a);                      \
    } while (0)
    if (g->dim_head == 64) { if (lo_mode) S3XB(64, true); else S3XB(64, false); }
    else { if (lo_mode) S3XB(32, true); else S3XB(32, false); }
#undef S3XB
    LAUNCH_CHECK();
    hipLaunchKernelGGL(s3_bwd_fin_kernel, dim3(g->B * (((int)inner + 63) / 64) + (g->heads * g->heads + 15) / 16), dim3(1024), 0, stream, a, g->dim_head);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

AMDNUWA_SAT_ACCESSOR(sparse3dna)
