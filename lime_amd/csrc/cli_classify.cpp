// cli_classify.cpp -- drop-in for the reference's `Classify` (src/Classify.cpp:304-751):
//   Classify N fileInput1 .. fileInputN numReads numGenomes fileOutput fileTaxo taxRank numThreads
// reads the N (2 or 4) .res files written by ClusterBWT_DA and writes the classification file.
// The reference's compile-time switches are environment variables here: LIME_BIN (default 1 =
// `make BIN=1`) and LIME_HIGHER (default 0 = `make HIGHER=0`).  numThreads is accepted and ignored
// (the output is in read order either way).
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <iostream>
#include <string>
#include <vector>

#include "lime_hip.h"

int main(int argc, char **argv)
{
    const auto t0 = std::chrono::steady_clock::now();
    const char *usage = " N fileInput1 fileInput2 ... fileInputN numReads numGenomes fileOutput fileTaxo taxRank numThreads";
    if (argc < 2) { std::cerr << "Error usage " << argv[0] << usage << std::endl; exit(1); }
    unsigned numFile = 0;
    sscanf(argv[1], "%u", &numFile);
    if (argc != (int)numFile + 8) { std::cerr << "Error usage " << argv[0] << usage << std::endl; exit(1); }
    if (numFile != 2 && numFile != 4) {
        std::cerr << "Error usage " << argv[0] << ": the allowed number of input files is 2 (single-end reads), or 4 (paired-end reads)" << std::endl;
        exit(1);
    }
    std::vector<const char *> in;
    for (unsigned i = 2; i < numFile + 2; ++i) in.push_back(argv[i]);
    unsigned numReads = 0, numTarg = 0;
    sscanf(argv[numFile + 2], "%u", &numReads);
    sscanf(argv[numFile + 3], "%u", &numTarg);
    const std::string fileOutput = argv[numFile + 4], fileTaxID = argv[numFile + 5];
    int taxRank = 0;
    sscanf(argv[numFile + 6], "%d", &taxRank);
    if (taxRank > 6 || taxRank < 0) {
        std::cerr << "Error usage: taxRank 0=Genome, 1=Species, 2=Genus, 3=Family, 4=Order, 5=Class, 6=Phylum." << std::endl;
        exit(1);
    }
    const char *eb = getenv("LIME_BIN"), *eh = getenv("LIME_HIGHER");
    const int BIN = eb ? atoi(eb) : 1, HIGHER = eh ? atoi(eh) : 0;
    std::cout << "Reading " << fileTaxID << std::endl;
    std::cout << "Reading files:";
    for (unsigned i = 0; i < numFile; ++i) {
        if (BIN) std::cout << "\n\t" << in[i] << ".bin" << "\n\t" << in[i] << ".pos";
        else std::cout << "\n\t" << in[i] << ".txt";
    }
    std::cout << std::endl;
    std::cerr << "Start comparing..." << std::endl;
    uint64_t counts[4];
    const int rc = lime_classify(numFile, in.data(), BIN, numReads, numTarg, fileOutput.c_str(), fileTaxID.c_str(), taxRank,
                                 HIGHER, counts);
    if (rc != LIME_OK) { std::cerr << lime_classify_error() << std::endl; exit(1); }
    std::cout << "Classification process at level " << taxRank << " completed.\nNumber of successfully classified reads: "
              << counts[0] << "/" << numReads << ";" << std::endl;
    if (HIGHER) std::cout << "\tClassified at higher taxonomic ranks: " << counts[3] << "." << std::endl;
    std::cout << "\tAmbiguously classified reads: " << counts[2] << "." << std::endl;
    std::cout << "\tNot classified reads: " << counts[1] << "." << std::endl;
    fprintf(stdout, "Time: %.6lf\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
