// What does leaving a process cost when it holds G GB of device memory (and P GB of page-locked host memory)?
//   exit_probe G P mode     mode: exit = _exit at once; free = hipFree / hipHostFree everything, then _exit; big = ONE allocation of G GB;
//                           pfree = the page-locked buffers freed by a thread each, then _exit
// prints the time of the allocations and of the frees; the caller times the whole process.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 100, P = argc > 2 ? atoi(argv[2]) : 1;
    const char *mode = argc > 3 ? argv[3] : "exit";
    const double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    const double t1 = now();
    std::vector<void *> d, h;
    const bool big = !strcmp(mode, "big") || !strcmp(mode, "bigfree");
    if (big) {
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)G << 30) != hipSuccess) return 1;
        hipMemset(p, 1, (size_t)G << 30);
        d.push_back(p);
    } else
        for (int k = 0; k < G; k++) {
            void *p = nullptr;
            if (hipMalloc(&p, (size_t)1 << 30) != hipSuccess) return 1;
            hipMemset(p, 1, (size_t)1 << 30);
            d.push_back(p);
        }
    for (int k = 0; k < P * 4; k++) {
        void *p = nullptr;
        if (hipHostMalloc(&p, (size_t)256 << 20, hipHostMallocDefault) != hipSuccess) return 1;
        memset(p, 1, (size_t)256 << 20);
        h.push_back(p);
    }
    // S streams (environment EXIT_PROBE_STREAMS), each used once: a hardware queue each
    const int S = getenv("EXIT_PROBE_STREAMS") ? atoi(getenv("EXIT_PROBE_STREAMS")) : 0;
    std::vector<hipStream_t> st((size_t)S);
    void *scratch = nullptr;
    if (S) hipMalloc(&scratch, 1 << 20);
    for (int k = 0; k < S; k++) {
        hipStreamCreateWithFlags(&st[(size_t)k], hipStreamNonBlocking);
        hipMemsetAsync(scratch, k, 1 << 20, st[(size_t)k]);
    }
    hipDeviceSynchronize();
    const double t2 = now();
    if (S && getenv("EXIT_PROBE_DESTROY"))
        for (auto x : st) hipStreamDestroy(x);
    if (!strcmp(mode, "pfree")) { // every page-locked buffer freed by a thread of its own
        std::vector<std::thread> th;
        for (void *p : h) th.emplace_back([p] { (void)hipHostFree(p); });
        for (auto &t : th) t.join();
    }
    if (!strcmp(mode, "free") || !strcmp(mode, "bigfree")) {
        for (void *p : d) hipFree(p);
        for (void *p : h) hipHostFree(p);
    }
    const double t3 = now();
    printf("init %.3f s, allocate+touch %.3f s, free %.3f s, in main %.3f s  ", t1 - t0, t2 - t1, t3 - t2, t3 - t0);
    fflush(stdout);
    _exit(0);
}
