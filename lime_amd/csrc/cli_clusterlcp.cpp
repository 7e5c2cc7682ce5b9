// Drop-in `ClusterLCP` (reference: src/ClusterLCP.cpp:47-320): same argv, same input files
// (fileFasta.lcp, fileFasta.da), same outputs (fileFasta.<alpha>.clrs, <base>.out).  The scan
// itself runs on the MI355X through lime_detect (include/lime_hip.h).  `threads` is accepted
// for command-line compatibility; records are always written in ascending pStart order
// (= the reference's 1-thread order; its multi-thread order is nondeterministic, :229-235).
#include <chrono>
#include <iostream>
#include <sstream>

#include "cli_common.h"

int main(int argc, char **argv)
{
    CliClock clk;
    if (argc != 6) {
        std::cerr << "Error usage: " << argv[0] << " fileFasta numReads numGenomes alpha threads" << std::endl;
        exit(1);
    }
    std::string fileFasta = argv[1];
    unsigned numReads = 0, numGenomes = 0, alpha = 0;
    int threads = 1;
    sscanf(argv[2], "%u", &numReads);
    sscanf(argv[3], "%u", &numGenomes);
    sscanf(argv[4], "%u", &alpha);
    sscanf(argv[5], "%d", &threads);
    io_threads_from_argv(threads);
    printf("Number of threads: %d (host); scan on GPU\n", threads);

    std::string fnLCP = fileFasta + ".lcp", fnDA = fileFasta + ".da";
    std::stringstream ss;
    ss << fileFasta << "." << alpha << ".clrs";
    const std::string fnOut = ss.str();

    MappedFile lcp, da;
    std::cout << "\n\t" << fnLCP;
    if (!lcp.open(fnLCP)) { std::cerr << "Error opening " << fnLCP << "." << std::endl; exit(EXIT_FAILURE); }
    std::cout << "\n\t" << fnDA << std::endl;
    if (!da.open(fnDA)) { std::cerr << "Error opening " << fnDA << "." << std::endl; exit(EXIT_FAILURE); }
    const uint64_t n = lcp.bytes / 4;
    if (da.bytes / 4 < n) { std::cerr << "Error: " << fnDA << " is shorter than " << fnLCP << "." << std::endl; exit(EXIT_FAILURE); }

    auto t0 = std::chrono::steady_clock::now();
    clk.mark("arguments, files mapped");
    lime_ctx *ctx = nullptr;
    if (lime_init(pick_device(), &ctx) != LIME_OK) { std::cerr << "Error: " << lime_last_error() << std::endl; exit(EXIT_FAILURE); }
    clk.mark("lime_init (HIP runtime)");
    uint64_t nClusters = 0, maxLen = 0;
    // the records go to the .clrs file chunk by chunk while the scan goes on (lime_detect_to_file)
    int rc = lime_detect_to_file(ctx, (const uint32_t *)lcp.data, (const uint32_t *)da.data, n, numReads, alpha, fnOut.c_str(), &nClusters, &maxLen);
    if (rc == LIME_ERR_IO) { std::cerr << "Error opening " << fnOut << "."; exit(1); }
    if (rc != LIME_OK) { std::cerr << "Error: " << lime_last_error() << std::endl; exit(EXIT_FAILURE); }
    clk.mark("scan + .clrs");
    const std::string fileaux = aux_name(fileFasta);
    if (lime_write_aux(fileaux.c_str(), numReads, numGenomes, alpha, maxLen, nClusters) != LIME_OK) {
        std::cerr << "Error opening " << fileaux << "." << std::endl; exit(EXIT_FAILURE);
    }
    lime_shutdown(ctx);
    clk.mark("aux file, shutdown");
    std::cout << "Clustering process with alpha=" << alpha << " completed.\nTotal number of clusters: " << nClusters
              << ".\nMaximum cluster size: " << maxLen << "." << std::endl;
    fprintf(stdout, "Time: %.6lf\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
