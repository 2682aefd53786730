// How fast can a file that sits in the page cache reach the device?  (the program's run is bound by
// page cache -> page-locked memory -> PCIe.)  Measures, with T threads over one file of N GB:
//   read      read() into a page-locked ring, no copy to the device            (the CPU copy alone)
//   dma       copies from page-locked memory to the device, no file            (the DMA alone)
//   read+dma  the program's path: read() into a page-locked ring + hipMemcpyAsync
//   mmap+reg  mmap the file, hipHostRegister a piece, hipMemcpyAsync, hipHostUnregister   (no CPU copy)
//   mmap      mmap the file, hipMemcpy from the pageable mapping
//   dma1 / read+dma1   as dma / read+dma with every thread's copies on ONE stream (what pjb_bam_piece did until round 6)
// build: hipcc -O2 -o /tmp/h2d_paths tools/debug/h2d_paths.cc -lpthread ; run: /tmp/h2d_paths FILE GB
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static const size_t PIECE = 64u << 20;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const char *path = argv[1];
    const size_t total = (size_t)(atof(argv[2]) * (1u << 30)) / PIECE * PIECE;
    {
        struct stat sb;
        if (stat(path, &sb) == 0 && (size_t)sb.st_size < total) {
            fprintf(stderr, "%s exists and is shorter than asked for: not touched\n", path);
            return 2;
        }
        if (stat(path, &sb) != 0) {
            int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
            std::vector<char> buf(PIECE);
            for (size_t i = 0; i < PIECE; i++) buf[i] = (char)(i * 2654435761u >> 13);
            for (size_t o = 0; o < total; o += PIECE)
                if (write(fd, buf.data(), PIECE) != (ssize_t)PIECE) return 3;
            close(fd);
        }
    }
    CK(hipSetDevice(0));
    void *dev;
    CK(hipMalloc(&dev, total));
    const size_t n_pieces = total / PIECE;
    const int RING = 3;
    const bool quick = argc > 3; // (a third argument: the program's path and its two halves only)
    std::vector<const char *> modes = {"read", "dma", "read+dma", "mmap+reg", "mmap", "read+dma", "mmap+reg"};
    if (quick) modes = {"dma1", "dma", "read+dma1", "read+dma", "read+dma1", "read+dma"};
    hipStream_t shared;
    CK(hipStreamCreateWithFlags(&shared, hipStreamNonBlocking));
    for (const char *mode : modes) {
        for (int T : {1, 2, 4, 8, 16}) {
            std::atomic<size_t> next{0};
            std::atomic<int> bad{0};
            std::vector<std::thread> th;
            std::vector<void *> pinned((size_t)T * RING);
            std::string m = mode;
            const bool one_stream = m.back() == '1';
            if (one_stream) m.pop_back();
            if (m != "mmap+reg" && m != "mmap")
                for (auto &p : pinned) CK(hipHostMalloc(&p, PIECE, hipHostMallocDefault));
            void *map = nullptr;
            int mfd = -1;
            if (m == "mmap+reg" || m == "mmap") {
                mfd = open(path, O_RDONLY);
                map = mmap(nullptr, total, PROT_READ, MAP_PRIVATE, mfd, 0);
                if (map == MAP_FAILED) return 4;
            }
            const double t0 = now();
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    CK(hipSetDevice(0));
                    hipStream_t st = shared;
                    if (!one_stream) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                    hipEvent_t ev[RING];
                    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    int fd = open(path, O_RDONLY);
                    size_t k = 0;
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i >= n_pieces) break;
                        const int slot = (int)(k++ % RING);
                        char *d = (char *)dev + i * PIECE;
                        if (m == "mmap+reg") {
                            char *src = (char *)map + i * PIECE;
                            if (hipHostRegister(src, PIECE, hipHostRegisterDefault) != hipSuccess) {
                                bad++;
                                (void)hipGetLastError();
                                break;
                            }
                            CK(hipMemcpyAsync(d, src, PIECE, hipMemcpyHostToDevice, st));
                            CK(hipStreamSynchronize(st));
                            CK(hipHostUnregister(src));
                            continue;
                        }
                        if (m == "mmap") {
                            CK(hipMemcpy(d, (char *)map + i * PIECE, PIECE, hipMemcpyHostToDevice));
                            continue;
                        }
                        void *p = pinned[(size_t)t * RING + slot];
                        if (k > (size_t)RING) CK(hipEventSynchronize(ev[slot]));
                        if (m != "dma") {
                            size_t got = 0;
                            while (got < PIECE) {
                                ssize_t r = pread(fd, (char *)p + got, PIECE - got, (off_t)(i * PIECE + got));
                                if (r <= 0) exit(5);
                                got += (size_t)r;
                            }
                        }
                        if (m != "read") {
                            CK(hipMemcpyAsync(d, p, PIECE, hipMemcpyHostToDevice, st));
                            CK(hipEventRecord(ev[slot], st));
                        }
                    }
                    CK(hipStreamSynchronize(st));
                    close(fd);
                    for (auto &e : ev) CK(hipEventDestroy(e));
                    if (!one_stream) CK(hipStreamDestroy(st));
                });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            printf("%-9s T=%2d: %6.2f GB/s (%.3f s)%s\n", mode, T, total / dt / 1e9, dt, bad ? "  [hipHostRegister refused]" : "");
            fflush(stdout);
            for (auto &p : pinned)
                if (p && m != "mmap+reg" && m != "mmap") CK(hipHostFree(p));
            if (map) munmap(map, total), close(mfd);
        }
    }
    CK(hipFree(dev));
    return 0;
}
