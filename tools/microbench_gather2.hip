// microbench_gather2.hip -- where does the random 64-byte gather fall off with table size, and does quad-cooperative
// loading (four lanes fetch the four 16-byte chunks of one lane's block, then a 4x4 register transpose with DPP)
// recover it while every lane still owns a whole block?
//   mode 0: lane loads its own block with four 16-B loads (current ps_core.h pattern)
//   mode 1: quad-cooperative: 4 instructions fetch the 4 blocks of a quad, 64 contiguous bytes per 4 lanes; transpose
// Each lane walks its own dependent chain.  Both modes XOR all 16 words so the result also checks the transpose.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned fold(uint4 a, uint4 b, uint4 c, uint4 d)
{
    return a.x ^ (a.y * 3u) ^ (a.z * 5u) ^ (a.w * 7u) ^ (b.x * 11u) ^ (b.y * 13u) ^ (b.z * 17u) ^ (b.w * 19u) ^
           (c.x * 23u) ^ (c.y * 29u) ^ (c.z * 31u) ^ (c.w * 37u) ^ (d.x * 41u) ^ (d.y * 43u) ^ (d.z * 47u) ^ (d.w * 53u);
}
__global__ void k_lane(const uint4 *tab, unsigned n_blocks, int steps, unsigned *out)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned idx = (tid * 2654435761u) % n_blocks, acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint4 *p = tab + (size_t)idx * 4;
        uint4 a = p[0], b = p[1], d = p[2], e = p[3];
        unsigned v = fold(a, b, d, e);
        acc += v;
        idx = (v * 2654435761u + idx) % n_blocks;
    }
    out[tid] = acc;
}
template <int CTRL> __device__ __forceinline__ unsigned dpp(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ uint4 dpp4(uint4 v) { return make_uint4(dpp<CTRL>(v.x), dpp<CTRL>(v.y), dpp<CTRL>(v.z), dpp<CTRL>(v.w)); }
__device__ __forceinline__ uint4 sel(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
__global__ void k_coop(const uint4 *tab, unsigned n_blocks, int steps, unsigned *out)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned q = threadIdx.x & 3;
    const bool q0 = (q & 1) != 0, q1 = (q & 2) != 0;
    unsigned idx = (tid * 2654435761u) % n_blocks, acc = 0;
    for (int s = 0; s < steps; ++s) {
        // chunk q of the block of quad lane j, j = 0..3
        const unsigned i0 = dpp<0x00>(idx), i1 = dpp<0x55>(idx), i2 = dpp<0xAA>(idx), i3 = dpp<0xFF>(idx);
        uint4 v0 = tab[(size_t)i0 * 4 + q], v1 = tab[(size_t)i1 * 4 + q], v2 = tab[(size_t)i2 * 4 + q], v3 = tab[(size_t)i3 * 4 + q];
        // 4x4 transpose inside the quad: stage 1 exchanges with lane^1, stage 2 with lane^2
        {
            const uint4 sa = sel(q0, v0, v1), sb = sel(q0, v2, v3);
            const uint4 ra = dpp4<0xB1>(sa), rb = dpp4<0xB1>(sb);
            v0 = sel(q0, ra, v0); v1 = sel(q0, v1, ra); v2 = sel(q0, rb, v2); v3 = sel(q0, v3, rb);
        }
        {
            const uint4 sa = sel(q1, v0, v2), sb = sel(q1, v1, v3);
            const uint4 ra = dpp4<0x4E>(sa), rb = dpp4<0x4E>(sb);
            v0 = sel(q1, ra, v0); v2 = sel(q1, v2, ra); v1 = sel(q1, rb, v1); v3 = sel(q1, v3, rb);
        }
        unsigned v = fold(v0, v1, v2, v3);
        acc += v;
        idx = (v * 2654435761u + idx) % n_blocks;
    }
    out[tid] = acc;
}

int main(int argc, char **argv)
{
    for (double gb : {0.5, 1.2, 1.6, 1.9, 2.07, 2.3, 3.0, 4.0}) {
        unsigned n_blocks = (unsigned)(gb * 1e9 / 64);
        uint4 *tab; unsigned *out;
        CK(hipMalloc(&tab, (size_t)n_blocks * 64));
        std::vector<unsigned> h((size_t)n_blocks * 16);
        unsigned x = 12345; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
        CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        int wpc = 12;
        int blocks = 256 * wpc / 4, steps = 300;
        CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        std::vector<unsigned> r0((size_t)blocks * 256), r1((size_t)blocks * 256);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, out);
                else hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy((mode ? r1 : r0).data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
            double loads = (double)blocks * 256 * steps;
            printf("table %.2f GB  waves/CU %2d  %-28s %7.2f ms  %6.2f G blocks/s  %7.1f GB/s\n", gb, wpc,
                   mode == 0 ? "lane-per-block" : "quad-cooperative + transpose", ms, loads / ms / 1e6, loads * 64 / ms / 1e6);
        }
        printf("   results identical: %s\n", r0 == r1 ? "yes" : "NO");
        CK(hipFree(out)); CK(hipFree(tab));
    }
    return 0;
}
