// microbench_store.hip -- what do scattered 16-byte stores (the stack pushes of the search kernel) cost the memory system?
//   mode 0: every lane stores ONE 16-byte entry into a random 64-byte line of a buffer far larger than the caches (partial line)
//   mode 1: every lane stores the four 16-byte quarters of a random line with four instructions (whole line, piecewise)
//   mode 2: four neighbouring lanes store the four quarters of ONE random line with one instruction (whole line, one request)
// Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: a read-for-ownership shows as FETCH_SIZE of a kernel that loads nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE>
__global__ void k_store(uint4 *buf, unsigned long long n_lines, int steps)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = (unsigned long long)(MODE == 2 ? tid >> 2 : tid) * 0x9E3779B97F4A7C15ull + 12345;
    for (int s = 0; s < steps; ++s) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const unsigned long long line = x % n_lines;
        const uint4 v = make_uint4((unsigned)x, (unsigned)(x >> 32), tid, (unsigned)s);
        if (MODE == 0) buf[line * 4 + (x >> 60 & 3)] = v;
        else if (MODE == 1) { buf[line * 4] = v; buf[line * 4 + 1] = v; buf[line * 4 + 2] = v; buf[line * 4 + 3] = v; }
        else buf[line * 4 + (tid & 3)] = v;
    }
}
int main()
{
    const size_t bytes = (size_t)16 << 30;                 // 16 GB: nothing of it stays in L2 / Infinity Cache between two touches
    uint4 *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    const unsigned long long n_lines = bytes / 64;
    const int blocks = 256 * 16, steps = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, buf, n_lines, steps);
            else if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, buf, n_lines, steps);
            else hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, buf, n_lines, steps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double lanes = (double)blocks * 256 * steps, lines = mode == 2 ? lanes / 4 : lanes;
        printf("mode %d: %8.2f ms  %7.2f G lines/s touched  %8.1f GB/s of entries stored\n", mode, ms, lines / ms / 1e6, lanes * (mode == 1 ? 64 : 16) / ms / 1e6);
    }
    return 0;
}
