// H2D copy rate of page-locked 64 MB pieces over 1 / 2 / 4 streams (hipMemcpyAsync), and of a kernel that reads the mapped host
// memory itself: how the file pieces of the end-to-end run should cross PCIe.   hipcc -O2 --offload-arch=gfx950 -o dma_probe dma_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void pull(const uint4 *src, uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t piece = 64u << 20, n_pieces = 96; // 6 GB
    {   // what allocations cost (first-time allocations are what the end-to-end run spends its first half second on)
        auto now0 = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        (void)hipFree(nullptr);
        for (size_t mb : {64, 256, 1024, 4096, 16384}) {
            void *q = nullptr;
            double t0 = now0();
            if (hipMalloc(&q, mb << 20) != hipSuccess) break;
            const double ta = now0() - t0;
            t0 = now0();
            (void)hipFree(q);
            printf("hipMalloc %5zu MB: %.1f ms, hipFree %.1f ms\n", mb, ta * 1e3, (now0() - t0) * 1e3);
        }
        for (size_t mb : {64, 256, 768}) {
            void *q = nullptr;
            double t0 = now0();
            if (hipHostMalloc(&q, mb << 20, hipHostMallocDefault) != hipSuccess) break;
            const double ta = now0() - t0;
            t0 = now0();
            (void)hipHostFree(q);
            printf("hipHostMalloc %5zu MB: %.1f ms, hipHostFree %.1f ms\n", mb, ta * 1e3, (now0() - t0) * 1e3);
        }
    }
    std::vector<void *> host(12);
    for (auto &h : host) {
        if (hipHostMalloc(&h, piece, hipHostMallocDefault) != hipSuccess) return 1;
        memset(h, 1, piece);
    }
    void *dev = nullptr;
    if (hipMalloc(&dev, piece * n_pieces) != hipSuccess) return 1;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int ns : {1, 2, 3, 4}) {
        std::vector<hipStream_t> st(ns);
        for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            for (size_t k = 0; k < n_pieces; k++)
                (void)hipMemcpyAsync((char *)dev + k * piece, host[k % host.size()], piece, hipMemcpyHostToDevice, st[k % ns]);
            for (auto &s : st) (void)hipStreamSynchronize(s);
            const double dt = now() - t0;
            if (rep) printf("hipMemcpyAsync, %d stream(s): %.1f GB/s\n", ns, piece * n_pieces / dt / 1e9);
        }
        for (auto &s : st) (void)hipStreamDestroy(s);
    }
    for (int blocks : {64, 256, 1024}) {
        hipStream_t s;
        (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            for (size_t k = 0; k < n_pieces; k++) {
                void *dp = nullptr;
                (void)hipHostGetDevicePointer(&dp, host[k % host.size()], 0);
                hipLaunchKernelGGL(pull, dim3(blocks), dim3(256), 0, s, (const uint4 *)dp, (uint4 *)((char *)dev + k * piece), piece / 16);
            }
            (void)hipStreamSynchronize(s);
            const double dt = now() - t0;
            if (rep) printf("kernel reading mapped host memory, %d blocks: %.1f GB/s\n", blocks, piece * n_pieces / dt / 1e9);
        }
        (void)hipStreamDestroy(s);
    }
    return 0;
}
