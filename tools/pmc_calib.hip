// FETCH_SIZE / WRITE_SIZE calibration kernels: known byte counts in the access patterns the libamdnuwa kernels use, so that the
// counter traffic quoted for a kernel family can be corrected by the factor measured for ITS pattern (MI355X_MICROARCH.md, HBM
// section: only the 16 B/lane streaming read is calibrated there, x2; "calibrate on a known byte count in your own access pattern").
//
//   stream16_read    lane l reads 16 contiguous bytes, lanes contiguous          (GEMM loaders, LayerNorm bf16 rows)
//   stream16_f32row  lane l reads a float4 of a 2 KiB fp32 row (LayerNorm rows)  (same coalescing, one row per wave pass)
//   gather128_read   8 lanes read one 128-byte row at a pseudo-random row index   (3DNA key / value rows, one (token, head) slice)
//   gather64_read    4 lanes read one 64-byte half row (the fragment-shaped loads of the first 3DNA score pass)
//   lds_dma16        global_load_lds_dwordx4: 64 lanes x 16 bytes straight to LDS (GEMM / cross-attention rings)
//   stream16_write   lane l writes 16 contiguous bytes; stream8_write: 8 bytes (bf16x4 epilogue / LN stores)
//   scatter128_write 8 lanes write one 128-byte row at a pseudo-random row index
// Buffers are 2 GiB (>> the 256 MiB Infinity Cache); every kernel touches each byte exactly once per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib && rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void sink(uint4 v, uint32_t* out) { if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345677u) out[0] = 1; }
__device__ __forceinline__ size_t perm(size_t i, size_t n) { return (i * 2654435761ull + 12345ull) % n; }   // n a power of two: odd multiplier = bijection

__global__ __launch_bounds__(256) void stream16_read(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    sink(acc, out);
}
__global__ __launch_bounds__(256) void stream16_f32row(const float4* __restrict__ p, size_t rows, uint32_t* out) {   // 2 KiB rows, one wave per row pass
    const int lane = threadIdx.x & 63; const size_t wv = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    float s = 0.f;
    for (size_t r = wv; r < rows; r += nw) { const float4 a = p[r * 128 + lane], b = p[r * 128 + 64 + lane]; s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w; }
    if (s == 1234.5678f) out[0] = 1;
}
template <int LANES>      // LANES x 16 bytes = one gathered piece (8 -> 128-byte rows, 4 -> 64-byte half rows)
__global__ __launch_bounds__(256) void gather_read(const uint4* __restrict__ p, size_t pieces, uint32_t* out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    for (size_t i = t; i < pieces * LANES; i += nt) { const uint4 v = p[perm(i / LANES, pieces) * LANES + (i % LANES)]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    sink(acc, out);
}
__global__ __launch_bounds__(256) void lds_dma16(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char tile[4][1024];
    const int wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) void* lds_p; typedef __attribute__((address_space(1))) const void* glb_p;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        __builtin_amdgcn_global_load_lds((glb_p)(p + i), (lds_p)tile[wave], 16, 0, 0);
    __syncthreads();
    if (reinterpret_cast<uint32_t*>(tile[wave])[threadIdx.x & 63] == 0x12345677u) out[0] = 1;
}
__global__ __launch_bounds__(256) void stream16_write(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void stream8_write(uint2* __restrict__ p, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) p[i] = make_uint2((uint32_t)i, 1);
}
__global__ __launch_bounds__(256) void scatter128_write(uint4* __restrict__ p, size_t rows) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    for (size_t i = t; i < rows * 8; i += nt) p[perm(i / 8, rows) * 8 + (i % 8)] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    void *buf = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 256)); CK(hipMemset(buf, 1, bytes)); CK(hipMemset(out, 0, 256));
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16_read, grid, block, 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(stream16_f32row, grid, block, 0, 0, (const float4*)buf, bytes / 2048, out);
        hipLaunchKernelGGL(gather_read<8>, grid, block, 0, 0, (const uint4*)buf, bytes / 128, out);
        hipLaunchKernelGGL(gather_read<4>, grid, block, 0, 0, (const uint4*)buf, bytes / 64, out);
        hipLaunchKernelGGL(lds_dma16, grid, block, 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(stream16_write, grid, block, 0, 0, (uint4*)buf, bytes / 16);
        hipLaunchKernelGGL(stream8_write, grid, block, 0, 0, (uint2*)buf, bytes / 8);
        hipLaunchKernelGGL(scatter128_write, grid, block, 0, 0, (uint4*)buf, bytes / 128);
    }
    CK(hipDeviceSynchronize());
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
