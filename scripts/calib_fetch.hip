// FETCH_SIZE calibration for the engine's access patterns (MI355X_MICROARCH.md: the counter
// is calibrated only for wide coalesced streams, where it reads 1/2 of the bytes).
//   k_stream : 16 B per lane, fully coalesced, 4 GiB once            -> known bytes = 4 GiB
//   k_rand16 : one random aligned 16-B slot per lane out of 4 GiB    -> 64-B sectors touched = N
//   k_rand8x8: one random 64-B-aligned run of 8 x 8 B per 8 lanes    -> 64-B sectors touched = N/8
// build: hipcc --offload-arch=gfx950 -O3 scripts/calib_fetch.hip -o gpurun_out/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t tmh(uint32_t x) { x = ((x >> 16) ^ x) * 0x45d9f3bu; x = ((x >> 16) ^ x) * 0x45d9f3bu; return (x >> 16) ^ x; }
__global__ void k_stream(const uint4* p, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_rand16(const uint4* p, uint32_t mask, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = p[tmh((uint32_t)i * 2654435761u + 17) & mask]; acc ^= v.x ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_rand8x8(const uint64_t* p, uint32_t mask64, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t grp = i >> 3;                       // 8 consecutive lanes share one random 64-B line
        uint64_t line = tmh((uint32_t)grp * 2654435761u + 5) & mask64;
        uint64_t v = p[line * 8 + (i & 7)]; acc ^= (uint32_t)v;
    }
    if (acc == 0x12345u) *sink = acc;
}
int main() {
    const uint64_t bytes = 4ull << 30;
    void* buf; uint32_t* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
    const uint64_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, n16, sink);
    const uint64_t nr = 1ull << 26;
    hipLaunchKernelGGL(k_rand16, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, (uint32_t)(n16 - 1), nr, sink);
    hipLaunchKernelGGL(k_rand8x8, dim3(8192), dim3(256), 0, 0, (const uint64_t*)buf, (uint32_t)(bytes / 64 - 1), nr, sink);
    hipDeviceSynchronize();
    printf("stream bytes %llu ; rand16 accesses %llu (x64 B = %llu) ; rand8x8 lines %llu (x64 B = %llu)\n",
           (unsigned long long)bytes, (unsigned long long)nr, (unsigned long long)nr * 64, (unsigned long long)(nr / 8), (unsigned long long)(nr / 8) * 64);
    return 0;
}
