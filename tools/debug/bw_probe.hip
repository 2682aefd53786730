// Streaming-read probe: what does the load width per lane do to achieved HBM bandwidth on this GPU?
// hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip && ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) 4 B per lane, IT loads per thread strided by the block (the shape of k1_count / k4a_simple)
template <int IT>
__global__ __launch_bounds__(256) void rd4(const int32_t *a, size_t n, int32_t *out) {
    size_t base = (size_t)blockIdx.x * 256 * IT;
    int32_t v[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) { size_t r = base + i * 256 + threadIdx.x; v[i] = r < n ? a[r] : 0; }
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < IT; i++) s += v[i];
    if (s == 0x7fffffff) out[0] = s;
}
// (b) 16 B per lane, IT loads per thread
template <int IT>
__global__ __launch_bounds__(256) void rd16(const int4 *a, size_t n4, int32_t *out) {
    size_t base = (size_t)blockIdx.x * 256 * IT;
    int4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) { size_t r = base + i * 256 + threadIdx.x; v[i] = r < n4 ? a[r] : make_int4(0, 0, 0, 0); }
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < IT; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 0x7fffffff) out[0] = s;
}
// (c) five arrays at once, 4 B per lane each (pos, cig_off, l_qseq, mtid, mpos) vs 16 B per lane each
__global__ __launch_bounds__(256) void rd5x4(const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *d, const int32_t *e, size_t n, int32_t *out) {
    size_t base = (size_t)blockIdx.x * 1024;
    int32_t s = 0;
    int32_t v[4][5];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        size_t r = base + i * 256 + threadIdx.x;
        bool on = r < n;
        v[i][0] = on ? a[r] : 0; v[i][1] = on ? b[r] : 0; v[i][2] = on ? c[r] : 0; v[i][3] = on ? d[r] : 0; v[i][4] = on ? e[r] : 0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s += v[i][0] + v[i][1] + v[i][2] + v[i][3] + v[i][4];
    if (s == 0x7fffffff) out[0] = s;
}
__global__ __launch_bounds__(256) void rd5x16(const int4 *a, const int4 *b, const int4 *c, const int4 *d, const int4 *e, size_t n4, int32_t *out) {
    size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n4) return;
    int4 x = a[r], y = b[r], z = c[r], w = d[r], u = e[r];
    int32_t s = x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w + z.x + z.y + z.z + z.w + w.x + w.y + w.z + w.w + u.x + u.y + u.z + u.w;
    if (s == 0x7fffffff) out[0] = s;
}

int main() {
    const size_t n = (size_t)64 << 20; // 64 M int32 = 256 MB per array
    int32_t *buf[5], *out;
    for (auto &p : buf) { CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 1, n * 4)); }
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, double bytes, auto launch) {
        for (int k = 0; k < 3; k++) launch();
        hipEventRecord(e0);
        const int R = 20;
        for (int k = 0; k < R; k++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us  %7.0f GB/s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e9);
    };
    for (size_t m : {(size_t)8 << 20, (size_t)64 << 20}) {
        printf("-- %zu M elements per array\n", m >> 20);
        time("4B/lane x1", m * 4.0, [&] { rd4<1><<<(m + 255) / 256, 256>>>(buf[0], m, out); });
        time("4B/lane x4", m * 4.0, [&] { rd4<4><<<(m + 1023) / 1024, 256>>>(buf[0], m, out); });
        time("4B/lane x8", m * 4.0, [&] { rd4<8><<<(m + 2047) / 2048, 256>>>(buf[0], m, out); });
        time("16B/lane x1", m * 4.0, [&] { rd16<1><<<(m / 4 + 255) / 256, 256>>>((const int4 *)buf[0], m / 4, out); });
        time("16B/lane x4", m * 4.0, [&] { rd16<4><<<(m / 4 + 1023) / 1024, 256>>>((const int4 *)buf[0], m / 4, out); });
        time("5 arrays 4B/lane x4", m * 20.0, [&] { rd5x4<<<(m + 1023) / 1024, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], m, out); });
        time("5 arrays 16B/lane x1", m * 20.0, [&] { rd5x16<<<(m / 4 + 255) / 256, 256>>>((const int4 *)buf[0], (const int4 *)buf[1], (const int4 *)buf[2], (const int4 *)buf[3], (const int4 *)buf[4], m / 4, out); });
    }
    return 0;
}
