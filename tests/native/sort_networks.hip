// sorting networks of mcq_device.hpp on the GPU against std::sort: the stage-by-stage forms, the one-block forms and the two-register forms
// (tests/test_gpu_sort_networks.py builds and runs this; exit code 0 = all equal)
#include <hip/hip_runtime.h>
#include "mcq_device.hpp"
using namespace mcq;
// out layout: 7 arrays of n u32 (one wave per 64 / 128 keys)
__global__ void k_sorts(const u32* in, u32* out, u32 n) {
    const u32 lane = threadIdx.x & 63;
    const u32 i = blockIdx.x * 64 + lane;
    u32 v = in[i];
    out[0 * n + i] = wave_sort64(v, lane);           // reference forms
    out[1 * n + i] = wave_sort64_1(v);
    out[2 * n + i] = wave_sort_blocks32(v);
    out[3 * n + i] = wave_sort_blocks32_1(v);
    u32 a = v, b = in[(i + 64) % n];
    u32 ra = wave_sort64(a, lane), rb = wave_sort64(b, lane);
    wave_sort64_x2(a, b);
    out[4 * n + i] = (a == ra && b == rb) ? 1u : 0u;
    // chain6 on a bitonic input: ascending in lanes 0..31, descending in 32..63 of sorted keys
    u32 s = wave_sort64(v, lane);
    u32 bit = __shfl(s, lane < 32 ? 2 * lane : 2 * (63 - lane) + 1, 64);
    u32 c0 = cx_j32(bit); c0 = cx_j16(c0); c0 = cx_j8(c0); c0 = cx_j4(c0); c0 = cx_j2(c0); c0 = cx_j1(c0);
    u32 c1 = cx_chain6_1(bit);
    u32 pa = bit, pb = ~bit;                         // (complemented: descending-ascending, still bitonic)
    u32 qa = pa, qb = pb; cx_chain6_x2(qa, qb);
    u32 eb = cx_j32(pb); eb = cx_j16(eb); eb = cx_j8(eb); eb = cx_j4(eb); eb = cx_j2(eb); eb = cx_j1(eb);
    out[5 * n + i] = (c0 == c1 && c0 == s) ? 1u : 0u;
    out[6 * n + i] = (qa == c0 && qb == eb) ? 1u : 0u;
    u32 r[4] = { in[(4 * (i / 64) * 64 + lane) % n], in[(4 * (i / 64) * 64 + 64 + lane) % n], in[(4 * (i / 64) * 64 + 128 + lane) % n], in[(4 * (i / 64) * 64 + 192 + lane) % n] };
    wave_regsort<u32, 4>(r, lane);
    bool ok = r[0] <= r[1] && r[1] <= r[2] && r[2] <= r[3];
    u32 nx = __shfl(r[0], (lane + 1) & 63, 64);
    ok = ok && (lane == 63 || r[0] <= nx);
    out[7 * n + i] = ok ? 1u : 0u;
}
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main() {
    const u32 nw = 1024, n = nw * 64;
    std::vector<u32> in(n);
    u32 x = 12345u;
    for (u32 i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        u32 v = x ^ (x >> 15);
        const u32 w = i / 64;
        if (w % 4 == 1) v &= 0xFFu;                   // many duplicates
        if (w % 4 == 2) v |= 0xFFFFFF00u;             // near the padding value
        if (w % 16 == 3 && (i & 7) == 0) v = 0xFFFFFFFFu;
        if (w % 16 == 7 && (i & 3) == 0) v = 0u;
        in[i] = v;
    }
    u32 *d_in, *d_out;
    CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_out, 8ull * n * 4));
    CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sorts, dim3(nw), dim3(64), 0, 0, d_in, d_out, n);
    CK(hipDeviceSynchronize());
    std::vector<u32> out(8ull * n);
    CK(hipMemcpy(out.data(), d_out, 8ull * n * 4, hipMemcpyDeviceToHost));
    u32 bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 w = 0; w < nw; ++w) {
        std::vector<u32> ref(in.begin() + w * 64, in.begin() + w * 64 + 64);
        std::vector<u32> lo(ref.begin(), ref.begin() + 32), hi(ref.begin() + 32, ref.end());
        std::sort(ref.begin(), ref.end()); std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
        for (u32 l = 0; l < 64; ++l) {
            const u32 i = w * 64 + l;
            if (out[0ull * n + i] != ref[l]) ++bad[0];
            if (out[1ull * n + i] != ref[l]) ++bad[1];
            const u32 b32 = l < 32 ? lo[l] : hi[l - 32];
            if (out[2ull * n + i] != b32) ++bad[2];
            if (out[3ull * n + i] != b32) ++bad[3];
            for (u32 k = 4; k < 8; ++k) if (out[(u64)k * n + i] != 1u) ++bad[k];
        }
    }
    const char* names[8] = {"wave_sort64 (stage-wise)", "wave_sort64_1", "wave_sort_blocks32 (stage-wise)", "wave_sort_blocks32_1", "wave_sort64_x2", "cx_chain6_1", "cx_chain6_x2", "wave_regsort<4>"};
    u32 total = 0;
    for (int k = 0; k < 8; ++k) { std::printf("%-34s %s (%u wrong lanes)\n", names[k], bad[k] ? "FAILED" : "ok", bad[k]); total += bad[k]; }
    return total ? 1 : 0;
}
