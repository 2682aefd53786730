// portcullis_amd: command line entry: the `junc` mode and `bamfilt`; prep / filt remain the
// reference's programs and interoperate through the prep directory and the .tab file.
#include <dirent.h>
#include <portcullis/bam_filter.hpp>
#include <portcullis/junction_builder.hpp>

#include <cerrno>
#include <csignal>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdlib>
#include <cstdio>
#include <ctime>
#include <iostream>

#ifndef PORTCULLIS_AMD_VERSION
#define PORTCULLIS_AMD_VERSION "1.2.4"
#endif

// PORTCULLIS_EARLY_RETURN=1 (opt-in): the work runs in a child process and this one returns as soon as the child says its
// outputs are written and closed.  Why one may want it: a process that held 100 GB of device memory takes the driver up
// to a second to tear down after it has nothing left to do, and whoever waits for the command waits for that too.  Why
// it is not the default: the child still holds its device memory while the driver tears it down, so a command started
// right behind this one shares the GPU with that for a moment -- `junc A && junc B` would overlap two working sets.
// The child is forked before anything touches the GPU (never if /dev/kfd is open already or a profiler is preloaded: a
// forked child must not inherit a started runtime); it reports its exit code through a pipe, closes its standard
// streams and leaves.
static pid_t g_child = -1;
static void forwardSignal(int sig) {
    if (g_child > 0) kill(g_child, sig);
}
static int g_report_fd = -1;  // (child) where the exit code goes

int main(int argc, char* argv[]) {
    // exit codes as in src/portcullis.cc:497-515 of the reference
    int rc = 0;
    // (never with the GPU runtime started in this process already -- a profiler's or any other tool's preloaded library may have
    // done that before main: decided from the process state, an open /dev/kfd, not from the tool's name)
    bool kfd_open = false;
    {
        // (every entry of /proc/self/fd, whatever its number; a /proc that cannot be read counts as "started": no fork then)
        DIR* d = opendir("/proc/self/fd");
        if (!d) kfd_open = true;
        else {
            char link[300], target[256];
            while (const dirent* e = readdir(d)) {
                if (e->d_name[0] == '.') continue;
                snprintf(link, sizeof link, "/proc/self/fd/%s", e->d_name);
                const ssize_t n = readlink(link, target, sizeof target - 1);
                if (n > 0) {
                    target[n] = 0;
                    if (strstr(target, "/dev/kfd") != nullptr || strstr(target, "/dev/dri/render") != nullptr) kfd_open = true;
                }
            }
            closedir(d);
        }
    }
    const char* early = getenv("PORTCULLIS_EARLY_RETURN");
    if (argc >= 2 && early && atoi(early) != 0 && !kfd_open && !getenv("HSA_TOOLS_LIB") && !getenv("PJB_NORMAL_EXIT")) {
        int fds[2];
        if (pipe(fds) == 0) {
            std::cout.flush();
            std::cerr.flush();
            const pid_t pid = fork();
            if (pid > 0) {  // the waiting side
                close(fds[1]);
                g_child = pid;
                signal(SIGINT, forwardSignal);
                signal(SIGTERM, forwardSignal);
                unsigned char code = 0;
                ssize_t got;
                do got = read(fds[0], &code, 1);
                while (got < 0 && errno == EINTR);
                if (got == 1) _exit((int)code);
                int status = 0;  // the child ended without a word: its status is ours
                while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
                }
                if (WIFEXITED(status)) _exit(WEXITSTATUS(status));
                if (WIFSIGNALED(status)) {
                    signal(WTERMSIG(status), SIG_DFL);
                    raise(WTERMSIG(status));
                }
                _exit(8);
            }
            if (pid == 0) {
                close(fds[0]);
                g_report_fd = fds[1];
                // (the stages may leave device contexts and page-locked buffers to the end of the process: it comes right
                // after the report, and nobody waits for it)
                setenv("PORTCULLIS_CHILD_LEAVES", "1", 1);
            } else {  // fork failed: one process
                close(fds[0]);
                close(fds[1]);
            }
        }
    }
    // The program keeps ~15 HIP streams busy at once (file pieces, four inflate streams, the service stream, two per queued
    // chain, rows); the runtime maps them onto 4 hardware queues unless told otherwise, and streams that share a queue wait
    // for each other.  8 queues: end to end 2.36 -> 2.18 s median of 7 (profiles/r03o_e2e_hw_queues.txt); at the round's last
    // tree 6 is as good or a little better end to end (1.85 against 1.87 - 1.88 s; 4: 1.85 - 1.91, 12: 1.88 - 2.02;
    // profiles/r03cq_e2e_hw_queues.txt), and the kernel chains alone -- the bench's step -- run 12.5 ms on 4 to 6 queues and
    // 14.3 on 8 (profiles/r03cp_hw_queues.txt).  Read by the runtime when it starts, so it is set before anything touches
    // HIP; a value the user exported wins.
    setenv("GPU_MAX_HW_QUEUES", "6", 0);
    if (getenv("PJB_PROFILE_HOST")) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "[host profile] main entered at epoch %.6f\n", (double)ts.tv_sec + ts.tv_nsec * 1e-9);
    }
    try {
        if (argc < 2 || (strcmp(argv[1], "junc") != 0 && strcmp(argv[1], "bamfilt") != 0)) {
            std::cerr << "Usage: portcullis_amd junc [options] <prep_data_dir>" << std::endl
                      << "       portcullis_amd bamfilt [options] <junction-file> <bam-file>" << std::endl;
            return 1;
        }
        portcullis::JunctionSystem::version = PORTCULLIS_AMD_VERSION;
        if (strcmp(argv[1], "bamfilt") == 0) rc = portcullis::BamFilter::main(argc - 1, argv + 1);
        else rc = portcullis::JunctionBuilder::main(argc - 1, argv + 1);
    } catch (const portcullis::PortcullisException& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        rc = 4;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        rc = 5;
    } catch (...) {
        std::cerr << "Error: Exception of unknown type!" << std::endl;
        rc = 7;
    }
    // every output file is written and closed: leave without unloading the HIP runtime (which may still be starting
    // on its own thread after an early error) and without walking the heap (0.15-0.2 s at process exit)
    std::cout.flush();
    std::cerr.flush();
    if (getenv("PJB_PROFILE_HOST")) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "[host profile] leaving main at epoch %.6f\n", (double)ts.tv_sec + ts.tv_nsec * 1e-9);
    }
    if (getenv("PJB_NORMAL_EXIT")) return rc;  // (profilers write their traces from exit handlers)
    if (g_report_fd >= 0) {  // every output is written and closed: tell the waiting side, let go of its terminal, leave
        const unsigned char code = (unsigned char)rc;
        ssize_t w;
        do w = write(g_report_fd, &code, 1);
        while (w < 0 && errno == EINTR);
        close(g_report_fd);
        close(0);
        close(1);
        close(2);
    }
    _exit(rc);
}
