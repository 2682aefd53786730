// How fast do N threads pread a page-cache-warm file into (a) ordinary memory, (b) page-locked memory (hipHostMalloc)?
// hipcc -O2 -o pread_probe pread_probe.cc -lpthread ; ./pread_probe <file> <threads>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double read_all(int fd, size_t size, char *dst, int nthreads, size_t chunk) {
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([=] {
            size_t per = (size + nthreads - 1) / nthreads, a = per * t, b = std::min(size, a + per);
            while (a < b) {
                size_t n = std::min(chunk, b - a);
                ssize_t r = pread(fd, dst + a, n, (off_t)a);
                if (r <= 0) break;
                a += (size_t)r;
            }
        });
    for (auto &x : th) x.join();
    return size / (now() - t0) / 1e9;
}
int main(int argc, char **argv) {
    const char *path = argv[1];
    int nthreads = argc > 2 ? atoi(argv[2]) : 16;
    int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    size_t size = (size_t)st.st_size;
    char *plain = (char *)malloc(size);
    memset(plain, 1, size);
    char *pinned = nullptr;
    if (hipHostMalloc((void **)&pinned, size, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    memset(pinned, 1, size);
    printf("file %.1f GB, %d threads\n", size / 1e9, nthreads);
    for (int rep = 0; rep < 2; rep++) {
        printf("  pread -> malloc   (1 MB calls): %6.1f GB/s\n", read_all(fd, size, plain, nthreads, 1 << 20));
        printf("  pread -> malloc  (64 MB calls): %6.1f GB/s\n", read_all(fd, size, plain, nthreads, 64 << 20));
        printf("  pread -> pinned  (64 MB calls): %6.1f GB/s\n", read_all(fd, size, pinned, nthreads, 64 << 20));
    }
    // memcpy from a mapping of the file
    char *m = (char *)mmap(nullptr, size, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
    if (m != MAP_FAILED) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
            th.emplace_back([=] { size_t per = (size + nthreads - 1) / nthreads, a = per * t, b = std::min(size, a + per); if (a < b) memcpy(pinned + a, m + a, b - a); });
        for (auto &x : th) x.join();
        printf("  memcpy mmap -> pinned         : %6.1f GB/s\n", size / (now() - t0) / 1e9);
    }
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([=] { size_t per = (size + nthreads - 1) / nthreads, a = per * t, b = std::min(size, a + per); if (a < b) memcpy(pinned + a, plain + a, b - a); });
    for (auto &x : th) x.join();
    printf("  memcpy malloc -> pinned       : %6.1f GB/s\n", size / (now() - t0) / 1e9);
    return 0;
}
