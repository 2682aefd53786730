// portcullis_amd: command line entry: the `junc` mode and `bamfilt`; prep / filt remain the
// reference's programs and interoperate through the prep directory and the .tab file.
#include <portcullis/bam_filter.hpp>
#include <portcullis/junction_builder.hpp>

#include <cstring>
#include <unistd.h>
#include <cstdlib>
#include <cstdio>
#include <ctime>
#include <iostream>

#ifndef PORTCULLIS_AMD_VERSION
#define PORTCULLIS_AMD_VERSION "1.2.4"
#endif

int main(int argc, char* argv[]) {
    // exit codes as in src/portcullis.cc:497-515 of the reference
    int rc = 0;
    // The program keeps ~15 HIP streams busy at once (file pieces, four inflate streams, the service stream, two per queued
    // chain, rows); the runtime maps them onto 4 hardware queues unless told otherwise, and streams that share a queue wait
    // for each other.  8 queues: end to end 2.36 -> 2.18 s median of 7 (profiles/r03o_e2e_hw_queues.txt).  Read by the
    // runtime when it starts, so it is set before anything touches HIP; a value the user exported wins.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
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
    _exit(rc);
}
