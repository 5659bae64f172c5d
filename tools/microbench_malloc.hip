// microbench_malloc.hip -- what does the search workspace cost to allocate?  (measurement aid, not a product path)
// ps_map's first search launch waited ~4 s for its 69 GB stack workspace.  Times hipMalloc / hipFree of several
// sizes, one block against many smaller ones, a second allocation right after a free, and the stream-ordered allocator.
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench_malloc tools/microbench_malloc.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(unsigned *p, size_t n_words, size_t stride_words) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride_words; if (i < n_words) p[i] = 1u; }
int main()
{
    double t0 = now();
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    std::printf("runtime + device initialised in %.3f s\n", now() - t0);
    const size_t GB = (size_t)1 << 30;
    for (size_t gb : {(size_t)1, (size_t)8, (size_t)32, (size_t)69}) {
        void *p = nullptr;
        double a = now(); CK(hipMalloc(&p, gb * GB)); double b = now();
        hipLaunchKernelGGL(touch, dim3((unsigned)(gb * GB / 4 / 524288 / 256 + 1)), dim3(256), 0, 0, (unsigned *)p, gb * GB / 4, (size_t)524288);   // one word per 2 MB
        CK(hipDeviceSynchronize()); double c = now();
        CK(hipFree(p)); double d = now();
        std::printf("hipMalloc %3zu GB: %.3f s, first touch of every 2 MB page %.3f s, hipFree %.3f s\n", gb, b - a, c - b, d - c);
    }
    {   // again, right after the free: does the driver keep anything?
        void *p = nullptr; double a = now(); CK(hipMalloc(&p, 69 * GB)); double b = now(); CK(hipFree(p));
        std::printf("hipMalloc 69 GB again: %.3f s\n", b - a);
    }
    {   // the same 69 GB as 69 blocks, and as 276 blocks of 256 MB
        for (size_t blk : {GB, GB / 4}) {
            std::vector<void *> v(69 * GB / blk);
            double a = now(); for (auto &p : v) CK(hipMalloc(&p, blk)); double b = now();
            for (auto &p : v) CK(hipFree(p));
            std::printf("69 GB as %zu blocks of %zu MB: %.3f s (free %.3f s)\n", v.size(), blk >> 20, b - a, now() - b);
        }
    }
    {   // stream-ordered allocator (pool keeps what is freed)
        hipStream_t s; CK(hipStreamCreate(&s));
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int rep = 0; rep < 2; ++rep) {
            void *p = nullptr; double a = now(); CK(hipMallocAsync(&p, 69 * GB, s)); CK(hipStreamSynchronize(s)); double b = now();
            CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s));
            std::printf("hipMallocAsync 69 GB (pass %d): %.3f s, hipFreeAsync %.3f s\n", rep, b - a, now() - b);
        }
    }
    {   // two threads' worth: 2 x 35 GB from the main thread one after the other
        void *p = nullptr, *q = nullptr; double a = now(); CK(hipMalloc(&p, 35 * GB)); double b = now(); CK(hipMalloc(&q, 34 * GB)); double c = now();
        std::printf("35 GB then 34 GB: %.3f + %.3f s\n", b - a, c - b); CK(hipFree(p)); CK(hipFree(q));
    }
    return 0;
}
