// microbench_gather3.hip -- dependent random gathers of 32-byte blocks (two 16-B loads per lane) against 64-byte blocks
// (four loads), by table size: decides whether an Occ layout of 64 symbols per 32-byte block (3.1 GB for hg19) stays on
// the good side of the table-size cliff seen between 3 and 4 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int CHUNKS>
__global__ void k_lane(const uint4 *tab, unsigned long long n_blocks, int steps, unsigned *out)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long idx = ((unsigned long long)tid * 2654435761ull) % n_blocks; unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint4 *p = tab + idx * CHUNKS;
        unsigned v = 0;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) { uint4 a = p[c]; v ^= a.x ^ (a.y * 3u) ^ (a.z * 5u) ^ (a.w * 7u); }
        acc += v;
        idx = ((unsigned long long)v * 2654435761ull + idx) % n_blocks;
    }
    out[tid] = acc;
}
__global__ void k_fill(unsigned *t, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        t[i] = (unsigned)x;
    }
}
int main()
{
    for (double gb : {2.07, 3.1, 3.3, 3.6, 4.0, 4.4}) {
        for (int chunks : {2, 4}) {
            const size_t bytes = (size_t)(gb * 1e9); const unsigned long long n_blocks = bytes / (16 * chunks);
            uint4 *tab; unsigned *out;
            CK(hipMalloc(&tab, bytes));
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *)tab, bytes / 4); CK(hipDeviceSynchronize());
            int wpc = 16, blocks = 256 * wpc / 4, steps = 300;
            CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (chunks == 2) hipLaunchKernelGGL(k_lane<2>, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, out);
                else hipLaunchKernelGGL(k_lane<4>, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double loads = (double)blocks * 256 * steps;
            printf("table %.2f GB  %2d-byte blocks  %7.2f ms  %6.2f G blocks/s  %7.1f GB/s\n", gb, 16 * chunks, ms, loads / ms / 1e6, loads * 16 * chunks / ms / 1e6);
            CK(hipFree(out)); CK(hipFree(tab));
        }
    }
    return 0;
}
