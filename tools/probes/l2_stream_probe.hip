// How fast can a CU pull an L2-resident stream?  256 workgroups x 8 waves; the 32 workgroups of an XCD (block b runs on XCD b % 8) read the
// SAME 512-KiB region over and over (the K / V images of one sample in the cross-attention kernels), 64 KiB per step:
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1-KiB pieces, two 64-KiB stages, vmcnt(0) + barrier per step)      -- what xattn6 does
//   mode 1: global_load_dwordx4 into VGPRs (8 per wave in flight), no LDS write
//   mode 2: global_load_dwordx4 + ds_write_b128 (register staging)
//   mode 3: as 0 without the barrier (vmcnt only)
// hipcc --offload-arch=gfx950 -O3 tools/probes/l2_stream_probe.hip -o /tmp/l2probe && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_vptr_t;
__device__ __forceinline__ void dma16_s(const char* sbase, unsigned voff, void* lds) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr_t)lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ buf, int region_bytes, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = buf + (size_t)(blockIdx.x & 7) * region_bytes;
    const int nst = region_bytes / 65536;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        const char* src = base + (size_t)(s % nst) * 65536;
        char* dst = smem + (s & 1) * 65536;
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dma16_s(src + (wave + 8 * i) * 1024, lane * 16, dst + (wave + 8 * i) * 1024);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (MODE == 0) __builtin_amdgcn_s_barrier();
        } else {
            uint4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const uint4*>(src + (wave + 8 * i) * 1024 + lane * 16);
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(dst + (wave + 8 * i) * 1024 + lane * 16) = r[i];
                __builtin_amdgcn_s_barrier();
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += r[i].x ^ r[i].w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2 || MODE == 0 || MODE == 3) { __syncthreads(); acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4); }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE>
float run(const char* buf, int region, int steps, unsigned* sink) {
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 131072, 0, buf, region, steps, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 131072, 0, buf, region, steps, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const int region = 512 * 1024, steps = 400;
    char* buf; unsigned* sink;
    hipMalloc(&buf, (size_t)8 * region); hipMemset(buf, 1, (size_t)8 * region); hipMalloc(&sink, 4);
    const double bytes = 256.0 * steps * 65536;
    for (int rep = 0; rep < 2; ++rep) {
        float t0 = run<0>(buf, region, steps, sink), t1 = run<1>(buf, region, steps, sink), t2 = run<2>(buf, region, steps, sink), t3 = run<3>(buf, region, steps, sink);
        printf("LDS-DMA + barrier %7.3f ms %6.2f TB/s | loads to VGPR %7.3f ms %6.2f TB/s | loads + ds_write + barrier %7.3f ms %6.2f TB/s | LDS-DMA no barrier %7.3f ms %6.2f TB/s\n",
               t0, bytes / t0 / 1e9, t1, bytes / t1 / 1e9, t2, bytes / t2 / 1e9, t3, bytes / t3 / 1e9);
    }
    return 0;
}
