// cli_egsatobcr.cpp -- drop-in for the reference's `EGSAtoBCR fastaFile numSeq` (src/EGSAtoBCR.cpp):
// splits eGSA's fastaFile.<numSeq>.gesa (13-byte records {u32 text, u32 suff, u32 lcp, u8 bwt}) into the
// three flat arrays the hot path reads: fastaFile.ebwt (u8), fastaFile.lcp (u32), fastaFile.da (u32 =
// the record's text index).  Only whole records count (a truncated tail is dropped, as the reference's
// feof loop does).  Host-only; block-wise I/O instead of one fread/fwrite per field.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc != 3) { std::cerr << "Error usage: " << argv[0] << " fastaFile numSeq" << std::endl; exit(1); }
    const std::string fasta = argv[1], fnEGSA = fasta + "." + argv[2] + ".gesa";
    FILE *in = fopen(fnEGSA.c_str(), "rb");
    if (!in) { std::cerr << "Error opening " << fnEGSA << std::endl; exit(EXIT_FAILURE); }
    std::cerr << "file EGSA: " << fnEGSA << "." << std::endl;
    const std::string fnBWT = fasta + ".ebwt", fnLCP = fasta + ".lcp", fnDA = fasta + ".da";
    FILE *ob = fopen(fnBWT.c_str(), "wb");
    if (!ob) { std::cerr << "Error opening " << fnBWT << "." << std::endl; exit(EXIT_FAILURE); }
    FILE *ol = fopen(fnLCP.c_str(), "wb");
    if (!ol) { std::cerr << "Error opening " << fnLCP << "." << std::endl; exit(EXIT_FAILURE); }
    FILE *od = fopen(fnDA.c_str(), "wb");
    if (!od) { std::cerr << "Error opening " << fnDA << "." << std::endl; exit(EXIT_FAILURE); }
    std::cerr << "file ebwt : " << fnBWT << "." << std::endl;
    std::cerr << "file lcp: " << fnLCP << "." << std::endl;
    std::cerr << "file da: " << fnDA << "." << std::endl;
    const size_t REC = 13, BATCH = 1 << 16;
    std::vector<uint8_t> buf(REC * BATCH), bw(BATCH);
    std::vector<uint32_t> lc(BATCH), da(BATCH);
    uint64_t total = 0;
    size_t carry = 0;                                   // bytes of a record split across two reads
    for (;;) {
        const size_t got = fread(buf.data() + carry, 1, buf.size() - carry, in) + carry;
        const size_t n = got / REC;
        for (size_t i = 0; i < n; ++i) {
            const uint8_t *r = buf.data() + i * REC;
            memcpy(&da[i], r, 4); memcpy(&lc[i], r + 8, 4); bw[i] = r[12];
        }
        if (n) {
            if (fwrite(bw.data(), 1, n, ob) != n || fwrite(lc.data(), 4, n, ol) != n || fwrite(da.data(), 4, n, od) != n) {
                std::cerr << "Error writing the output arrays." << std::endl; exit(EXIT_FAILURE);
            }
            total += n;
        }
        carry = got - n * REC;
        if (carry) memmove(buf.data(), buf.data() + n * REC, carry);
        if (got < buf.size()) break;                    // short read: end of file (a partial record is dropped)
    }
    fclose(in);
    if (fclose(ob) || fclose(ol) || fclose(od)) { std::cerr << "Error closing the output arrays." << std::endl; exit(EXIT_FAILURE); }
    std::cerr << "The total number of elements is " << total << "\n";
    return 0;
}
