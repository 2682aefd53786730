// Can the file pieces of the end-to-end run cross PCIe straight out of the page cache?  mmap a file that is in the page
// cache, hipHostRegister 64 MB pieces of the mapping, copy them to the device, unregister -- against pread into page-locked
// buffers + copy (what the program does).  T threads each.   hipcc -O2 --offload-arch=gfx950 -o register_probe register_probe.hip -lpthread
//   register_probe <file> [threads]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int T = argc > 2 ? atoi(argv[2]) : 8;
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) return 1;
    struct stat st;
    fstat(fd, &st);
    const size_t piece = 64u << 20, total = ((size_t)st.st_size / piece) * piece;
    if (!total) return 1;
    uint8_t *map = (uint8_t *)mmap(nullptr, total, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) return 1;
    (void)hipFree(nullptr);
    void *dev = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess) return 1;
    const size_t np = total / piece;
    {   // anonymous memory, registered: what a ring of page-locked buffers costs to set up, and how fast it then crosses
        const size_t nb = 12;
        double t0 = now();
        uint8_t *anon = (uint8_t *)mmap(nullptr, nb * piece, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(anon, nb * piece, MADV_HUGEPAGE);
        const double t_map = now() - t0;
        t0 = now();
        {
            std::vector<std::thread> th;
            for (int t = 0; t < 4; t++) th.emplace_back([&, t] { for (size_t k = t; k < nb; k += 4) memset(anon + k * piece, 1, piece); });
            for (auto &x : th) x.join();
        }
        const double t_touch = now() - t0;
        t0 = now();
        for (size_t k = 0; k < nb; k++)
            if (hipHostRegister(anon + k * piece, piece, hipHostRegisterDefault) != hipSuccess) printf("register failed\n");
        const double t_reg = now() - t0;
        hipStream_t s;
        (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        double best = 0;
        for (int rep = 0; rep < 3; rep++) {
            t0 = now();
            for (size_t k = 0; k < 48; k++) (void)hipMemcpyAsync((char *)dev + (k % np) * piece, anon + (k % nb) * piece, piece, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            best = std::max(best, 48.0 * piece / (now() - t0) / 1e9);
        }
        printf("anonymous ring of 12 x 64 MB: mmap %.1f ms, first touch by 4 threads %.1f ms, hipHostRegister %.1f ms (all 12); copies from it %.1f GB/s\n", t_map * 1e3,
               t_touch * 1e3, t_reg * 1e3, best);
        void *hm = nullptr;
        t0 = now();
        (void)hipHostMalloc(&hm, nb * piece, hipHostMallocDefault);
        const double t_hm = now() - t0;
        best = 0;
        for (int rep = 0; rep < 3; rep++) {
            t0 = now();
            for (size_t k = 0; k < 48; k++) (void)hipMemcpyAsync((char *)dev + (k % np) * piece, (char *)hm + (k % nb) * piece, piece, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            best = std::max(best, 48.0 * piece / (now() - t0) / 1e9);
        }
        printf("hipHostMalloc of the same 768 MB: %.1f ms; copies from it %.1f GB/s\n", t_hm * 1e3, best);
        for (size_t k = 0; k < nb; k++) (void)hipHostUnregister(anon + k * piece);
        (void)hipHostFree(hm);
        munmap(anon, nb * piece);
        (void)hipStreamDestroy(s);
        fflush(stdout);
    }
    for (int mode = 0; mode < 5; mode++) {
        if (mode == 1 || mode == 2) continue; // (measured before: profiles/r03ap_register_probe.txt)
        // 3: memcpy out of the mapping into page-locked buffers + copy; 4: the same with MADV_SEQUENTIAL + MADV_WILLNEED first
        // 0: pread into page-locked buffers + copy; 1: register the mapping's pieces + copy + unregister; 2: copy from the pageable mapping
        std::atomic<size_t> next(0);
        std::atomic<int> bad(0);
        double t_reg = 0;
        const double t0 = now();
        std::vector<std::thread> th;
        std::vector<double> regs((size_t)T, 0.0);
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                hipStream_t s;
                (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                void *pin = nullptr;
                if ((mode == 0 || mode >= 3) && hipHostMalloc(&pin, piece, hipHostMallocDefault) != hipSuccess) bad = 1;
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= np || bad) break;
                    uint8_t *src = map + k * piece;
                    if (mode == 0) {
                        size_t got = 0;
                        while (got < piece) {
                            const ssize_t r = pread(fd, (char *)pin + got, piece - got, (off_t)(k * piece + got));
                            if (r <= 0) { bad = 1; break; }
                            got += (size_t)r;
                        }
                        (void)hipMemcpyAsync((char *)dev + k * piece, pin, piece, hipMemcpyHostToDevice, s);
                        (void)hipStreamSynchronize(s);
                    } else if (mode >= 3) {
                        if (mode == 4) madvise(src, piece, MADV_WILLNEED);
                        memcpy(pin, src, piece);
                        (void)hipMemcpyAsync((char *)dev + k * piece, pin, piece, hipMemcpyHostToDevice, s);
                        (void)hipStreamSynchronize(s);
                    } else if (mode == 1) {
                        const double r0 = now();
                        if (hipHostRegister(src, piece, hipHostRegisterDefault) != hipSuccess) { bad = 2; break; }
                        regs[(size_t)t] += now() - r0;
                        (void)hipMemcpyAsync((char *)dev + k * piece, src, piece, hipMemcpyHostToDevice, s);
                        (void)hipStreamSynchronize(s);
                        (void)hipHostUnregister(src);
                    } else {
                        (void)hipMemcpyAsync((char *)dev + k * piece, src, piece, hipMemcpyHostToDevice, s);
                        (void)hipStreamSynchronize(s);
                    }
                }
                if (pin) (void)hipHostFree(pin);
                (void)hipStreamDestroy(s);
            });
        for (auto &x : th) x.join();
        const double dt = now() - t0;
        for (double r : regs) t_reg += r;
        const char *names[] = {"pread into page-locked buffers + copy", "hipHostRegister(mapping) + copy + unregister", "copy from the pageable mapping",
                               "memcpy from the mapping into page-locked buffers + copy", "the same after MADV_WILLNEED"};
        printf("%-48s %2d threads: %6.1f GB/s%s", names[mode], T, total / dt / 1e9, bad ? "  (FAILED)" : "");
        if (mode == 1) printf("   (register: %.1f ms per 64 MB piece)", t_reg / np * 1e3);
        printf("\n");
        fflush(stdout);
        (void)hipGetLastError();
    }
    return 0;
}
