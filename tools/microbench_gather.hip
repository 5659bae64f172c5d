// microbench_gather.hip -- ceiling for the FM-index access pattern on MI355X: dependent random 64-byte
// block gathers from a table much larger than L2 / Infinity Cache.
//   mode 0: one lane loads a whole 64-B block (4 x 16 B), the layout of ps_core.h (one read per lane)
//   mode 1: four lanes share one block (1 x 16 B each), a "quad per read" layout
// Each lane walks a dependent chain (next index derived from the loaded data), like a backward search.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_lane(const uint4 *tab, unsigned n_blocks, int steps, unsigned *out, int chains)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned idx[2], acc = 0;
    idx[0] = (tid * 2654435761u) % n_blocks; idx[1] = (tid * 40503u + 12345u) % n_blocks;
    for (int s = 0; s < steps; ++s) {
        for (int c = 0; c < chains; ++c) {
            const uint4 *p = tab + (size_t)idx[c] * 4;
            uint4 a = p[0], b = p[1], d = p[2], e = p[3];
            unsigned v = a.x ^ b.y ^ d.z ^ e.w;
            acc += v;
            idx[c] = (v * 2654435761u + idx[c]) % n_blocks;
        }
    }
    out[tid] = acc;
}
__global__ void k_quad(const uint4 *tab, unsigned n_blocks, int steps, unsigned *out)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, q = tid >> 2, sub = tid & 3;
    unsigned idx = (q * 2654435761u) % n_blocks, acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 a = tab[(size_t)idx * 4 + sub];
        unsigned v = a.x ^ a.y ^ a.z ^ a.w;
        v ^= __shfl_xor((int)v, 1, 64); v ^= __shfl_xor((int)v, 2, 64);
        acc += v;
        idx = (v * 2654435761u + idx) % n_blocks;
    }
    out[tid] = acc;
}

int main(int argc, char **argv)
{
    for (double gb : {0.23, 1.2, 4.0}) {
        unsigned n_blocks = (unsigned)(gb * 1e9 / 64);
        uint4 *tab; unsigned *out;
        CK(hipMalloc(&tab, (size_t)n_blocks * 64));
        std::vector<unsigned> h((size_t)n_blocks * 16);
        unsigned x = 12345; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
        CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int wpc : {8, 16, 32}) {                 // waves per CU resident
            int blocks = 256 * wpc / 4, steps = 400;
            CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, out, 1);
                    else if (mode == 1) hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps / 2, out, 2);
                    else hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double loads = mode == 2 ? (double)blocks * 64 * steps : (double)blocks * 256 * steps;
                printf("table %.2f GB  waves/CU %2d  %-26s %7.2f ms  %6.2f G blocks/s  %7.1f GB/s (64 B per block)\n", gb, wpc,
                       mode == 0 ? "lane-per-block, 1 chain" : mode == 1 ? "lane-per-block, 2 chains" : "quad-per-block", ms, loads / ms / 1e6, loads * 64 / ms / 1e6);
            }
            CK(hipFree(out));
        }
        CK(hipFree(tab));
    }
    return 0;
}
