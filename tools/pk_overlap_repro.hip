// Micro-kernel for the hypothesis of DESIGN.md 5q: does  v_pk_fma_f32 D, A, D, C op_sel:[0,1,0]  (destination pair == a source pair, the LOW
// result reading the source's HIGH register) ever produce a low half computed from the already-written high result when LDS read data is
// returning into the register file at the same time?  Every thread loops: two LDS reads (the second still in flight, as in the head-mix
// loop), the packed FMA on the pair the first read delivered, compare with the value computed from a second copy by plain code.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_overlap_repro.hip -o tools/pk_overlap_repro.co && ./tools/pk_overlap_repro.co
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void repro(unsigned* counters, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* L = reinterpret_cast<float*>(smem);
    const int t = threadIdx.x;
    const int nfl = 12288;                                   // 48 KiB of table
    for (int e = t; e < nfl; e += blockDim.x) L[e] = 0.25f + 0.001f * (float)((e * 37 + blockIdx.x) & 1023);
    __syncthreads();
    unsigned bad_lo = 0, bad_hi = 0;
    const f32x2 W = {1.5f + 0.01f * (t & 7), -0.75f + 0.02f * (t & 3)};
    const f32x2 C = {0.125f * (t & 15), 3.0f - 0.0625f * (t & 31)};
    for (int it = 0; it < iters; ++it) {
        const int idx = ((t * 8 + it * 4104) % (nfl - 16)) & ~3;           // 16-byte aligned, bank-conflicting on purpose
        const unsigned addr = (unsigned)(idx * 4);
        const float p1 = L[idx + 1];                          // the operand both halves must read
        f32x2 P; f32x4 junk;
        if (mode == 0) {
            asm volatile("ds_read_b64 %0, %2\n\t"
                         "ds_read_b128 %1, %2 offset:16\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         "v_pk_fma_f32 %0, %3, %0, %4 op_sel:[0,1,0]\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(P), "=&v"(junk) : "v"(addr), "v"(W), "v"(C) : "memory");
        } else {                                              // control: the same arithmetic into a DIFFERENT destination pair
            f32x2 D;
            asm volatile("ds_read_b64 %0, %3\n\t"
                         "ds_read_b128 %1, %3 offset:16\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         "v_pk_fma_f32 %2, %4, %0, %5 op_sel:[0,1,0]\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(P), "=&v"(junk), "=&v"(D) : "v"(addr), "v"(W), "v"(C) : "memory");
            P = D;
        }
        const float elo = __builtin_fmaf(W.x, p1, C.x), ehi = __builtin_fmaf(W.y, p1, C.y);
        bad_lo += (P.x != elo) ? 1u : 0u;
        bad_hi += (P.y != ehi) ? 1u : 0u;
        if (junk.x == 12345.678f) L[idx] = junk.y;            // keep the second read alive
    }
    if (bad_lo) atomicAdd(&counters[0], bad_lo);
    if (bad_hi) atomicAdd(&counters[1], bad_hi);
}

int main() {
    unsigned* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) { printf("no device\n"); return 2; }
    for (int mode = 0; mode < 2; ++mode)
        for (int lds_kib : {64, 128}) {                       // 64 KiB: two workgroups per CU; 128 KiB: one
            (void)hipMemset(d, 0, 8);
            (void)hipFuncSetAttribute((const void*)repro, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024);
            hipLaunchKernelGGL(repro, dim3(256 * 8), dim3(512), lds_kib * 1024, 0, d, 4000, mode);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 3; }
            unsigned h[2];
            (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
            printf("%s, %3d KiB LDS per workgroup (%s per CU): wrong low halves %u, wrong high halves %u of %lld\n",
                   mode == 0 ? "destination == source pair    " : "control: separate destination", lds_kib, lds_kib == 64 ? "two" : "one", h[0], h[1],
                   256LL * 8 * 512 * 4000);
        }
    return 0;
}
