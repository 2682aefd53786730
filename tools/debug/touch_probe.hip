// First touch of fresh device memory: hipMalloc returns at once on this box (0.2 ms for 16 GB) -- is the memory mapped when it is
// first written?  Times the first and the second pass of (a) hipMemsetAsync, (b) an H2D copy from page-locked memory, (c) a kernel
// that writes, over freshly allocated buffers.     hipcc -O2 --offload-arch=gfx950 -o touch_probe touch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void fill(uint4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t gb = 4, bytes = gb << 30;
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *host = nullptr;
    (void)hipHostMalloc(&host, 256u << 20, hipHostMallocDefault);
    memset(host, 1, 256u << 20);
    for (int mode = 0; mode < 3; mode++) {
        void *d = nullptr;
        double t0 = now();
        if (hipMalloc(&d, bytes) != hipSuccess) return 1;
        const double ta = now() - t0;
        double tt[2];
        for (int pass = 0; pass < 2; pass++) {
            t0 = now();
            if (mode == 0) (void)hipMemsetAsync(d, 0, bytes, s);
            else if (mode == 1)
                for (size_t off = 0; off < bytes; off += 256u << 20) (void)hipMemcpyAsync((char *)d + off, host, 256u << 20, hipMemcpyHostToDevice, s);
            else hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s, (uint4 *)d, bytes / 16);
            (void)hipStreamSynchronize(s);
            tt[pass] = now() - t0;
        }
        const char *names[3] = {"hipMemsetAsync", "H2D copy", "writing kernel"};
        printf("%-16s %zu GB: hipMalloc %.2f ms, first pass %.1f ms (%.1f GB/s), second pass %.1f ms (%.1f GB/s)\n", names[mode], gb, ta * 1e3, tt[0] * 1e3,
               gb * 1.073741824 / tt[0], tt[1] * 1e3, gb * 1.073741824 / tt[1]);
        t0 = now();
        (void)hipFree(d);
        printf("                 hipFree %.2f ms\n", (now() - t0) * 1e3);
    }
    return 0;
}
