// What limits "page cache -> page-locked ring -> device"?  G reader groups of T threads each read a page-cache-warm
// file in pieces into a ring of page-locked buffers (pread), optionally DMA every piece to the device, optionally
// while B other threads burn CPU (the FASTA readers).  Prints GB/s and the mean time of a piece read.
// hipcc -O2 -o ring_probe ring_probe.cc -lpthread ; ./ring_probe <file> <GB> <G> <T> <dma 0/1> <piece MB> <burners> [registered 0/1]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
__global__ void spin_kernel(long long cycles, float *buf, size_t n, int stream) {
    const long long t0 = clock64();
    float acc = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (clock64() - t0 < cycles) {
        if (stream) {
            acc += buf[i % n];
            i += (size_t)gridDim.x * blockDim.x;
        }
    }
    if (acc == 12345.f) buf[0] = acc;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    const size_t gb = (size_t)atoi(argv[2]);
    const int G = atoi(argv[3]), T = atoi(argv[4]), dma = atoi(argv[5]);
    const size_t piece = (size_t)atoi(argv[6]) << 20;
    const int burners = argc > 7 ? atoi(argv[7]) : 0;
    const int registered = argc > 8 ? atoi(argv[8]) : 0;
    const int gpu_busy = argc > 9 ? atoi(argv[9]) : 0;   // 1: a kernel that keeps every CU busy runs meanwhile; 2: it also streams through HBM
    const int churn = argc > 10 ? atoi(argv[10]) : 0;    // 1: another thread keeps allocating and freeing device memory
    const size_t size = gb << 30;
    struct stat st;
    if (stat(path, &st) != 0 || (size_t)st.st_size < size) {
        int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        std::vector<char> blk(64 << 20);
        for (size_t i = 0; i < blk.size(); i++) blk[i] = (char)(i * 2654435761u >> 13);
        for (size_t off = 0; off < size; off += blk.size()) (void)!write(fd, blk.data(), blk.size());
        close(fd);
    }
    const int NB = 3;
    std::vector<char *> bufs((size_t)G * NB);
    for (auto &b : bufs) {
        if (registered) {
            b = (char *)aligned_alloc(1 << 21, piece);
            memset(b, 1, piece);
            if (hipHostRegister(b, piece, hipHostRegisterDefault) != hipSuccess) { printf("hipHostRegister failed\n"); return 1; }
        } else {
            if (hipHostMalloc((void **)&b, piece, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
            memset(b, 1, piece);
        }
    }
    char *dev = nullptr;
    if (dma && hipMalloc((void **)&dev, size) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    std::atomic<bool> stop(false);
    std::vector<std::thread> burn;
    for (int b = 0; b < burners; b++) burn.emplace_back([&] { volatile double x = 1; while (!stop) for (int i = 0; i < 100000; i++) x = x * 1.0000001 + 1e-9; });
    std::thread gpu_thread, churn_thread;
    if (gpu_busy) gpu_thread = std::thread([&] {
        float *hb = nullptr;
        (void)hipMalloc((void **)&hb, 1ull << 30);
        hipStream_t ks;
        (void)hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
        while (!stop) {
            spin_kernel<<<2048, 256, 0, ks>>>(100000000ll, hb, (1ull << 30) / 4, gpu_busy == 2);  // ~50 ms per launch
            (void)hipStreamSynchronize(ks);
        }
    });
    if (churn) churn_thread = std::thread([&] {
        while (!stop) {
            void *p = nullptr;
            (void)hipMalloc(&p, 2ull << 30);
            (void)hipMemset(p, 0, 1 << 20);
            (void)hipFree(p);
        }
    });
    std::atomic<long> piece_us(0), pieces(0);
    const double t0 = now();
    std::vector<std::thread> groups;
    for (int g = 0; g < G; g++)
        groups.emplace_back([&, g] {
            int fd = open(path, O_RDONLY);
            hipStream_t s;
            (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            hipEvent_t ev[NB];
            for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            const size_t per = size / G, a0 = per * g, a1 = a0 + per;
            int k = 0;
            for (size_t off = a0; off < a1; off += piece, k++) {
                const size_t n = std::min(piece, a1 - off);
                char *buf = bufs[(size_t)g * NB + k % NB];
                if (dma && k >= NB) (void)hipEventSynchronize(ev[k % NB]);
                const double ta = now();
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++)
                    th.emplace_back([=] {
                        size_t sl = (n + T - 1) / T, a = sl * t, b = std::min(n, a + sl);
                        while (a < b) {
                            ssize_t r = pread(fd, buf + a, b - a, (off_t)(off + a));
                            if (r <= 0) break;
                            a += (size_t)r;
                        }
                    });
                for (auto &x : th) x.join();
                piece_us += (long)((now() - ta) * 1e6);
                pieces++;
                if (dma) {
                    (void)hipMemcpyAsync(dev + off, buf, n, hipMemcpyHostToDevice, s);
                    (void)hipEventRecord(ev[k % NB], s);
                }
            }
            if (dma) (void)hipStreamSynchronize(s);
            close(fd);
        });
    for (auto &x : groups) x.join();
    const double dt = now() - t0;
    stop = true;
    for (auto &x : burn) x.join();
    if (gpu_thread.joinable()) gpu_thread.join();
    if (churn_thread.joinable()) churn_thread.join();
    printf("gpu_busy=%d churn=%d G=%d T=%d dma=%d piece=%zuMB burners=%d registered=%d: %.1f GB in %.3f s = %.1f GB/s; mean piece read %.1f ms (%.1f GB/s per group)\n", gpu_busy, churn, G, T, dma,
           piece >> 20, burners, registered, size / 1e9, dt, size / dt / 1e9, piece_us / 1e3 / pieces, piece / (piece_us / 1e6 / pieces) / 1e9);
    return 0;
}
