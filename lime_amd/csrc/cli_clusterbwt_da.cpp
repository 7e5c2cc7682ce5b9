// Drop-in `ClusterBWT_DA` (reference: src/ClusterBWT_DA.cpp:453-773): same argv, same inputs
// (<base>.out, fileFasta.<alpha>.clrs, fileFasta.da, fileFasta.ebwt), same outputs
// (fileFasta.res.bin + .res.pos, or fileFasta.res.txt).  The reference's compile-time
// switches are runtime here: LIME_EBWT (default 1, Makefile:13) and LIME_BIN (default 1,
// Makefile:12).  Scoring and the row scan run on the MI355X through lime_score_choose.
#include <chrono>
#include <iostream>
#include <sstream>
#include <vector>

#include "cli_common.h"

static int env_flag(const char *name, int dflt)
{
    const char *s = getenv(name);
    return s ? atoi(s) != 0 : dflt;
}

int main(int argc, char **argv)
{
    CliClock clk;
    if (argc != 5) {
        std::cerr << "Error usage " << argv[0] << " fileFasta readLen beta threads" << std::endl;
        exit(1);
    }
    const int EBWT = env_flag("LIME_EBWT", 1), BIN = env_flag("LIME_BIN", 1);
    int threads = 1;
    sscanf(argv[4], "%d", &threads);
    io_threads_from_argv(threads);
    printf("Number of threads: %d (host); scoring on GPU\n", threads);
    std::string fileFasta = argv[1];
    unsigned char readLen = 0;                 // dataTypeSim, parsed with %hhu (:519-521)
    float beta = 0;
    sscanf(argv[2], "%hhu", &readLen);
    sscanf(argv[3], "%f", &beta);

    uint32_t numRead = 0, numRef = 0, minLCP = 0;
    uint64_t maxLen = 0, nClusters = 0;
    const std::string fileaux = aux_name(fileFasta);
    if (lime_read_aux(fileaux.c_str(), &numRead, &numRef, &minLCP, &maxLen, &nClusters) != LIME_OK) {
        std::cerr << "Error opening " << fileaux << "." << std::endl; exit(EXIT_FAILURE);
    }
    std::cout << "numRead: " << numRead << ", numRef: " << numRef << ", minLCP: " << minLCP
              << ", nClusters: " << nClusters << std::endl;
    const uint32_t norm = (uint32_t)(readLen + 1 - minLCP);    // :555
    if (maxLen > LIME_MAX_CLUSTER) {                            // :558-562
        std::cerr << "Error Usage: maximum cluster size is " << maxLen
                  << " greater than sizeMaxBuf, please increase sizeMaxBuf in Tools.h" << std::endl;
        exit(1);
    }
    std::stringstream ss;
    ss << fileFasta << "." << minLCP << ".clrs";
    const std::string fnCluster = ss.str(), fnDA = fileFasta + ".da", fnBWT = fileFasta + ".ebwt";
    MappedFile clrs, da, bwt;
    if (!clrs.open(fnCluster)) { std::cerr << "Error opening " << fnCluster << "." << std::endl; exit(EXIT_FAILURE); }
    if (!da.open(fnDA)) { std::cerr << "Error opening " << fnDA << "." << std::endl; exit(EXIT_FAILURE); }
    if (EBWT && !bwt.open(fnBWT)) { std::cerr << "Error opening " << fnBWT << "." << std::endl; exit(EXIT_FAILURE); }
    if (clrs.bytes / sizeof(lime_cluster_t) < nClusters) nClusters = clrs.bytes / sizeof(lime_cluster_t);
    const uint64_t n = da.bytes / 4;

    auto t0 = std::chrono::steady_clock::now();
    clk.mark("arguments, files mapped");
    std::cerr << "Computing similarity arrays SimArray_i[1,numRead]..." << std::endl;
    // the table stays in HBM: the row scan and the (idRef, sim) lists of the passing reads are made
    // on the device and only those come back (lime_score_choose).  LIME_GPUS=k: the cluster list is cut over k GPUs
    // of this process, one RCCL reduce-scatter of the tables (lime_score_choose_multi)
    std::vector<uint8_t> rmax((size_t)numRead + 1);
    std::vector<uint64_t> roff((size_t)numRead + 2);
    lime_pair_t *pairs = nullptr; uint64_t nPairs = 0;
    const int gpus = getenv("LIME_GPUS") ? atoi(getenv("LIME_GPUS")) : 0;
    lime_ctx *ctx = nullptr;
    int rc;
    if (gpus >= 1) {
        rc = lime_score_choose_multi(gpus, nullptr, (const uint32_t *)da.data, EBWT ? (const uint8_t *)bwt.data : nullptr, n,
                                     (const lime_cluster_t *)clrs.data, nClusters, numRead, numRef, norm, beta,
                                     rmax.data(), roff.data(), &pairs, &nPairs);
    } else {
        if (lime_init(pick_device(), &ctx) != LIME_OK) { std::cerr << "Error: " << lime_last_error() << std::endl; exit(EXIT_FAILURE); }
        clk.mark("lime_init (HIP runtime)");
        rc = lime_score_choose(ctx, (const uint32_t *)da.data, EBWT ? (const uint8_t *)bwt.data : nullptr, n,
                               (const lime_cluster_t *)clrs.data, nClusters, numRead, numRef, norm, beta,
                               rmax.data(), roff.data(), &pairs, &nPairs, nullptr);
    }
    if (rc != LIME_OK) { std::cerr << "Error: " << lime_last_error() << std::endl; exit(1); }
    fprintf(stderr, "TIME clusterAnalyze: %.6lf\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

    clk.mark("scoring + choose");
    auto t1 = std::chrono::steady_clock::now();
    const std::string fnF = fileFasta + ".res";
    if (BIN) {
        std::cerr << "Writing " << fnF << ".pos" << std::endl << "Writing " << fnF << ".bin" << std::endl;
        rc = lime_write_res_bin_pairs((fnF + ".bin").c_str(), (fnF + ".pos").c_str(), rmax.data(), roff.data(), pairs, numRead, norm, beta);
    } else {
        std::cerr << "Writing " << fnF << ".txt" << std::endl;
        rc = lime_write_res_txt_pairs((fnF + ".txt").c_str(), rmax.data(), roff.data(), pairs, numRead, norm, beta);
    }
    lime_free(pairs);
    if (rc != LIME_OK) { std::cerr << "Error opening " << fnF << "." << std::endl; exit(EXIT_FAILURE); }
    fprintf(stdout, "Time: %.6lf\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    clk.mark("result files");
    if (ctx) lime_shutdown(ctx);
    clk.mark("shutdown");
    std::cout << "Cluster analysis completed with beta=" << beta << "." << std::endl;
    std::cout << "Number of clusters: " << nClusters << "." << std::endl;
    fprintf(stdout, "Time: %.6lf\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
