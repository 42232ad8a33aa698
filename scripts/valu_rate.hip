// VALU issue-rate probe on MI355X: how many wave64 integer VALU instructions per cycle per SIMD?
// 8 waves/SIMD resident (2048 blocks x 256 threads), each wave runs ITER x 8 independent int ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = a ^ 9, g = a + 11, h = a * 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // plain add/xor
            a += b; b ^= c; c += d; d ^= e; e += f; f ^= g; g += h; h ^= a;
        } else if (MODE == 1) { // compare + select
            a = a < b ? c : a; b = b < c ? d : b; c = c < d ? e : c; d = d < e ? f : d;
            e = e < f ? g : e; f = f < g ? h : f; g = g < h ? a : g; h = h < a ? b : h;
        } else if (MODE == 2) { // 32-bit multiply
            a *= b; b *= c; c *= d; d *= e; e *= f; f *= g; g *= h; h *= a;
        } else {                // DPP move + xor
            a ^= (uint32_t)__builtin_amdgcn_update_dpp((int)a, (int)b, 0xB1, 0xF, 0xF, false);
            b ^= (uint32_t)__builtin_amdgcn_update_dpp((int)b, (int)c, 0x4E, 0xF, 0xF, false);
            c ^= (uint32_t)__builtin_amdgcn_update_dpp((int)c, (int)d, 0xB1, 0xF, 0xF, false);
            d ^= (uint32_t)__builtin_amdgcn_update_dpp((int)d, (int)e, 0x4E, 0xF, 0xF, false);
            e ^= (uint32_t)__builtin_amdgcn_update_dpp((int)e, (int)f, 0xB1, 0xF, 0xF, false);
            f ^= (uint32_t)__builtin_amdgcn_update_dpp((int)f, (int)g, 0x4E, 0xF, 0xF, false);
            g ^= (uint32_t)__builtin_amdgcn_update_dpp((int)g, (int)h, 0xB1, 0xF, 0xF, false);
            h ^= (uint32_t)__builtin_amdgcn_update_dpp((int)h, (int)a, 0x4E, 0xF, 0xF, false);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int MODE> void run(const char* name, uint32_t* out, int ops_per_iter) {
    const int iters = 20000, grid = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_insts = (double)grid * 4 * iters * ops_per_iter;          // per-wave VALU instructions
    double per_simd_per_s = wave_insts / 1024 / (ms * 1e-3);
    printf("%-18s %.2f ms  %.2f G wave-instr/s/SIMD  -> %.2f cycles per instr per SIMD at 2.4 GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
    uint32_t* out; hipMalloc(&out, 2048 * 256 * 4);
    run<0>("add/xor", out, 8); run<1>("cmp+cndmask (x2)", out, 16); run<2>("mul_lo_u32", out, 8); run<3>("dpp mov + xor (x2)", out, 16);
    return 0;
}
