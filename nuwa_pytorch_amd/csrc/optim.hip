// Optimiser step of the trainer (reference train_nuwa.py:253-255, optimizer.py:6-31):
// global-norm gradient clipping + AdamW over every parameter tensor in ONE pair of launches ("multi-tensor apply": a device table
// of <= 65536-element chunks, each pointing into one parameter / gradient / moment tensor).  HBM-bound: 16 B read + 12 B written per
// element.  The norm is a fixed-order two-stage reduction (deterministic) and stays on the device: the update kernel reads the clip
// coefficient from memory, so a step never synchronises with the host.
#include "common.h"
#include "../../include/amdnuwa.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const amdnuwa_adamw_chunk* __restrict__ ch, float* __restrict__ partials) {
    __shared__ float red[4];
    const amdnuwa_adamw_chunk c = ch[blockIdx.x];
    float s = 0.f;
    if (c.g) {
        const long long n4 = c.n & ~3LL;
        for (long long i = (long long)threadIdx.x * 4; i < n4; i += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(c.g + i);
            s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
        }
        for (long long i = n4 + threadIdx.x; i < c.n; i += 256) s += c.g[i] * c.g[i];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(1024) void grad_norm_finalize_kernel(const float* __restrict__ partials, int n, float max_norm, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0]);
        out[0] = norm;
        out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    }
}

__global__ __launch_bounds__(256) void scale_grads_kernel(const amdnuwa_adamw_chunk* __restrict__ ch, const float* __restrict__ coef) {
    const amdnuwa_adamw_chunk c = ch[blockIdx.x];
    const float k = coef[0];
    if (!c.g || k == 1.f) return;
    float* g = const_cast<float*>(c.g);
    for (long long i = threadIdx.x; i < c.n; i += 256) g[i] *= k;
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad):  p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
// p -= (lr / bias_c1) * m / (sqrt(v) / sqrt(bias_c2) + eps);   g is first multiplied by the clip coefficient
__global__ __launch_bounds__(256) void adamw_kernel(const amdnuwa_adamw_chunk* __restrict__ ch, float lr, float b1, float b2, float eps,
                                                    const float* __restrict__ clip) {
    const amdnuwa_adamw_chunk c = ch[blockIdx.x];
    if (!c.g) return;
    const float k = clip ? clip[0] : 1.f;
    const float decay = 1.f - lr * c.weight_decay, step = lr / c.bias_correction1, sqrt_bias_c2 = sqrtf(c.bias_correction2);
    auto upd = [&](float& p, float g, float& m, float& v) {
        g *= k;
        p *= decay;
        m = m + (1.f - b1) * (g - m);                          // lerp, as torch's exp_avg.lerp_(grad, 1 - beta1)
        v = b2 * v + (1.f - b2) * g * g;
        p -= step * (m / (sqrtf(v) / sqrt_bias_c2 + eps));     // addcdiv_(exp_avg, sqrt(v) / sqrt(bc2) + eps, value = -lr / bc1)
    };
    const long long n4 = c.n & ~3LL;
    for (long long i = (long long)threadIdx.x * 4; i < n4; i += 1024) {
        float4 p = *reinterpret_cast<float4*>(c.p + i), m = *reinterpret_cast<float4*>(c.m + i), v = *reinterpret_cast<float4*>(c.v + i);
        const float4 g = *reinterpret_cast<const float4*>(c.g + i);
        upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
        *reinterpret_cast<float4*>(c.p + i) = p; *reinterpret_cast<float4*>(c.m + i) = m; *reinterpret_cast<float4*>(c.v + i) = v;
    }
    for (long long i = n4 + threadIdx.x; i < c.n; i += 256) upd(c.p[i], c.g[i], c.m[i], c.v[i]);
}

}  // namespace

extern "C" int amdnuwa_grad_norm(const amdnuwa_adamw_chunk* chunks, int nchunks, float max_norm, float* partials, float* out2,
                                 hipStream_t stream) {
    if (!chunks || !partials || !out2 || nchunks < 0) return AMDNUWA_ERR_ARG;
    if (nchunks > 0) {
        hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, partials);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, nchunks, max_norm, out2);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_scale_grads(const amdnuwa_adamw_chunk* chunks, int nchunks, const float* coef, hipStream_t stream) {
    if (!chunks || !coef) return AMDNUWA_ERR_ARG;
    if (nchunks <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(scale_grads_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, coef);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}

extern "C" int amdnuwa_adamw_step(const amdnuwa_adamw_chunk* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                                  const float* clip_coef, hipStream_t stream) {
    if (!chunks) return AMDNUWA_ERR_ARG;
    if (nchunks <= 0) return AMDNUWA_OK;
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, lr, beta1, beta2, eps, clip_coef);
    LAUNCH_CHECK();
    return AMDNUWA_OK;
}
