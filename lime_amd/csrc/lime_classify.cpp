// lime_classify.cpp -- the read-assignment step that consumes the hot path's .res files
// (reference: src/Classify.cpp).  Host code: the work is a few float compares per
// read over lists the GPU already compacted.  Written from the reference's behaviour, not its text:
// every read's lists are first loaded into one in-memory form (from .res.bin/.res.pos or .res.txt),
// then one routine decides.  Decision rules and their reference lines:
//   U  no file holds a record for the read                                   (Classify.cpp:503-509)
//   1  files whose maximum is within ERROR of the best; their genomes within ERROR of the file's
//      maximum form the candidate set; one taxon at the chosen rank -> C     (:511-547)
//   2  else compare the candidates' per-strand sums (files 0+3 and 1+2 for paired-end): if one
//      strand wins by more than ERROR and its best candidates share one taxon -> C (:549-640)
//   3  else the same sums over ALL genomes; genomes within ERROR of the best: one taxon -> C,
//      otherwise (HIGHER) the lowest higher rank they share -> H, or A        (:642-690, :168-302)
// float arithmetic, the 0.02 tolerance and the text/binary difference (text values are the %.5f
// roundings) are kept as they are.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "lime_hip.h"

namespace {

const float TOL = static_cast<float>(0.02);      // ERROR, src/Tools.h:37
const int N_RANKS = 6;                           // species .. phylum (RANK, Classify.cpp:24)

struct Cell { float sim; uint32_t ref; };
struct ReadLists {                               // one read in one .res file
    float top = 0.0f;                            // the file's maximum for the read (0: no record)
    std::vector<Cell> cells;
    void clear() { top = 0.0f; cells.clear(); }
};

struct Taxonomy {
    std::vector<uint32_t> at_rank;               // genome -> taxon at the chosen rank (rank 0: genome index)
    std::vector<std::vector<uint32_t>> higher;   // [rank-1 .. 5][genome], 0 = unknown (HIGHER)
    std::string rank_name;
};

// ';'-separated lineage file, first line a header (Classify.cpp:32-85).  As in the reference a last
// line without a newline is not taken, and an empty field at the chosen rank is skipped.
bool read_taxonomy(const std::string &path, int rank, bool higher, uint32_t n_targ, Taxonomy &tx, std::string &err)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) { err = "Error opening file " + path + "."; return false; }
    std::string col[7];
    auto next_line = [&]() {
        for (int s = 0; s < 6; ++s) std::getline(f, col[s], ';');
        std::getline(f, col[6], '\n');
    };
    next_line();
    tx.rank_name = col[rank];
    if (higher) tx.higher.assign(N_RANKS, std::vector<uint32_t>(n_targ, 0u));
    uint32_t line = 0;
    next_line();
    while (f.good()) {
        if (rank > 0) {
            if (!col[rank].empty()) tx.at_rank.push_back((uint32_t)atoi(col[rank].c_str()));
        } else {
            tx.at_rank.push_back(line);
        }
        if (higher && line < n_targ)
            for (int i = (rank > 0 ? rank - 1 : 0); i < N_RANKS; ++i)
                if (!col[i + 1].empty()) tx.higher[i][line] = (uint32_t)atoi(col[i + 1].c_str());
        ++line;
        next_line();
    }
    return true;
}

// ---- sources of the per-read lists ---------------------------------------------------------
struct BinSource {                               // .res.bin + .res.pos (ClusterBWT_DA.cpp:376-436)
    FILE *bin = nullptr, *pos = nullptr;
    ~BinSource() { if (bin) fclose(bin); if (pos) fclose(pos); }
    bool open(const std::string &base) { bin = fopen((base + ".bin").c_str(), "rb"); pos = fopen((base + ".pos").c_str(), "rb"); return bin && pos; }
    bool read(uint64_t r, ReadLists &out)
    {
        out.clear();
        uint64_t p = 0;
        if (fseeko(pos, (off_t)(r * 8u), SEEK_SET) != 0 || fread(&p, 8, 1, pos) != 1) return false;
        if (!p) return true;
        struct { float sim; uint32_t id; } rec;
        if (fseeko(bin, (off_t)(p * 8u), SEEK_SET) != 0 || fread(&rec, 8, 1, bin) != 1) return false;
        out.top = rec.sim;
        const uint32_t n = rec.id;
        out.cells.resize(n);
        for (uint32_t k = 0; k < n; ++k) {
            if (fread(&rec, 8, 1, bin) != 1) return false;
            out.cells[k].sim = rec.sim; out.cells[k].ref = rec.id;
        }
        return true;
    }
};

struct TxtSource {                               // .res.txt, one line per read (ClusterBWT_DA.cpp:414-441)
    std::ifstream f;
    bool open(const std::string &base) { f.open((base + ".txt").c_str()); return f.is_open(); }
    bool read(uint64_t, ReadLists &out)
    {
        out.clear();
        std::string line;
        std::getline(f, line);
        std::istringstream is(line);
        if (!(is >> out.top)) { out.top = 0.0f; return true; }
        for (;;) {
            Cell c;
            if (!(is >> c.ref)) break;
            if (!(is >> c.sim)) c.sim = 0.0f;
            out.cells.push_back(c);
        }
        return true;
    }
};

struct Verdict { char type; uint32_t taxon; float sim; };

// value of genome g in a read's list (first match), 0 if absent
float value_of(const ReadLists &l, uint32_t g)
{
    for (const Cell &c : l.cells) if (c.ref == g) return c.sim;
    return 0.0f;
}

Verdict decide(const ReadLists *L, uint32_t n_files, uint32_t n_targ, const Taxonomy &tx, int rank, bool higher,
               std::vector<float> (&all)[2])
{
    // which files hold the read, and the best file maximum
    float best = 0.0f;
    bool any = false;
    for (uint32_t i = 0; i < n_files; ++i)
        if (L[i].top) { if (!any || L[i].top > best) best = L[i].top; any = true; }
    if (!any) return Verdict{'U', 0u, 0.0f};

    // rule 1: candidates = genomes close to the maximum of every file that is close to the best
    std::vector<uint32_t> cand;
    for (uint32_t i = 0; i < n_files; ++i) {
        if (!L[i].top || !(best - L[i].top < TOL)) continue;
        for (const Cell &c : L[i].cells) {
            if (!(L[i].top - c.sim < TOL)) continue;
            bool seen = false;
            for (uint32_t g : cand) if (g == c.ref) { seen = true; break; }
            if (!seen) cand.push_back(c.ref);
        }
    }
    std::set<uint32_t> taxa;
    for (uint32_t g : cand) taxa.insert(tx.at_rank[g]);
    if (taxa.size() == 1) return Verdict{'C', *taxa.begin(), best};

    // rule 2: per-strand sums of the candidates
    {
        float top2[2] = {0.0f, 0.0f};
        std::vector<float> s0(cand.size()), s1(cand.size());
        for (size_t e = 0; e < cand.size(); ++e) {
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t i = 0; i < n_files; ++i) v[i] = value_of(L[i], cand[e]);
            s0[e] = n_files == 4 ? v[0] + v[3] : v[0];
            s1[e] = n_files == 4 ? v[1] + v[2] : v[1];
            if (top2[0] < s0[e]) top2[0] = s0[e];
            if (top2[1] < s1[e]) top2[1] = s1[e];
        }
        const std::vector<float> *win = nullptr;
        float wtop = 0.0f;
        if (top2[0] > top2[1] + TOL) { win = &s0; wtop = top2[0]; }
        else if (top2[1] > top2[0] + TOL) { win = &s1; wtop = top2[1]; }
        if (win) {
            taxa.clear();
            for (size_t e = 0; e < cand.size(); ++e) if ((*win)[e] == wtop) taxa.insert(tx.at_rank[cand[e]]);
            if (taxa.size() == 1) return Verdict{'C', *taxa.begin(), wtop};
        }
    }

    // rule 3: the same sums over all genomes
    for (int k = 0; k < 2; ++k) all[k].assign(n_targ, 0.0f);
    {
        std::vector<float> f2, f3;
        if (n_files == 4) { f2.assign(n_targ, 0.0f); f3.assign(n_targ, 0.0f); }
        for (uint32_t i = 0; i < n_files; ++i)
            for (const Cell &c : L[i].cells) {
                if (c.ref >= n_targ) continue;
                if (i == 0) all[0][c.ref] += c.sim;
                else if (i == 1) all[1][c.ref] += c.sim;
                else if (i == 2) f2[c.ref] += c.sim;
                else f3[c.ref] += c.sim;
            }
        if (n_files == 4) for (uint32_t j = 0; j < n_targ; ++j) { all[0][j] += f3[j]; all[1][j] += f2[j]; }
    }
    float hi[2] = {0.0f, 0.0f};
    for (int k = 0; k < 2; ++k) for (uint32_t j = 0; j < n_targ; ++j) if (hi[k] < all[k][j]) hi[k] = all[k][j];
    std::vector<uint32_t> gens;                  // ascending genome index (the reference's std::set order)
    float h;
    if (hi[0] > hi[1]) { h = hi[0]; for (uint32_t j = 0; j < n_targ; ++j) if (h - all[0][j] < TOL) gens.push_back(j); }
    else if (hi[0] < hi[1]) { h = hi[1]; for (uint32_t j = 0; j < n_targ; ++j) if (h - all[1][j] < TOL) gens.push_back(j); }
    else { h = hi[0]; for (uint32_t j = 0; j < n_targ; ++j) if ((h - all[0][j] < TOL) || (h - all[1][j] < TOL)) gens.push_back(j); }
    if (gens.empty()) return Verdict{'A', 0u, 0.0f};
    bool one = true;
    for (uint32_t g : gens) if (tx.at_rank[g] != tx.at_rank[gens[0]]) one = false;
    if (one) return Verdict{'C', tx.at_rank[gens[0]], h};
    if (higher && rank >= 1) {
        for (int idx = rank - 1; idx < N_RANKS; ++idx) {
            const uint32_t t = tx.higher[idx][gens[0]];
            bool same = true;
            for (uint32_t g : gens) if (tx.higher[idx][g] != t) { same = false; break; }
            if (same && t != 0u) return Verdict{'H', t, h};
        }
    }
    return Verdict{'A', 0u, 0.0f};
}

template <typename Source>
int run(uint32_t n_files, const char *const *inputs, uint32_t n_reads, uint32_t n_targ, const Taxonomy &tx, int rank,
        bool higher, std::ofstream &out, uint64_t counts[4], std::string &err)
{
    std::vector<Source> src(n_files);
    for (uint32_t i = 0; i < n_files; ++i)
        if (!src[i].open(inputs[i])) { err = std::string("Error opening ") + inputs[i]; return LIME_ERR_IO; }
    ReadLists L[4];
    std::vector<float> all[2];
    for (uint32_t r = 0; r < n_reads; ++r) {
        for (uint32_t i = 0; i < n_files; ++i)
            if (!src[i].read(r, L[i])) { err = std::string("Error reading ") + inputs[i]; return LIME_ERR_IO; }
        for (uint32_t i = 0; i < n_files; ++i)
            for (const Cell &c : L[i].cells)
                if (c.ref >= n_targ) { err = std::string("genome index beyond numGenomes in ") + inputs[i]; return LIME_ERR_ARG; }
        const Verdict v = decide(L, n_files, n_targ, tx, rank, higher, all);
        switch (v.type) {
        case 'U': out << "U," << r << ",NA,0\n"; ++counts[1]; break;
        case 'A': out << "A," << r << ",NA,0\n"; ++counts[2]; break;
        case 'C': out << "C," << r << "," << v.taxon << "," << v.sim << "\n"; ++counts[0]; break;
        default:  out << "H," << r << "," << v.taxon << "," << v.sim << "\n"; ++counts[3]; break;
        }
    }
    return LIME_OK;
}

thread_local std::string g_cls_err;

} // namespace

extern "C" const char *lime_classify_error(void) { return g_cls_err.c_str(); }

extern "C" int lime_classify(uint32_t n_files, const char *const *inputs, int binary, uint32_t n_reads, uint32_t n_targ,
                             const char *path_out, const char *path_tax, int rank, int higher, uint64_t counts[4])
{
    uint64_t local[4] = {0, 0, 0, 0};
    if (!counts) counts = local;
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if ((n_files != 2 && n_files != 4) || !inputs || !path_out || !path_tax || rank < 0 || rank > N_RANKS) {
        g_cls_err = "lime_classify: bad argument";
        return LIME_ERR_ARG;
    }
    Taxonomy tx;
    if (!read_taxonomy(path_tax, rank, higher != 0, n_targ, tx, g_cls_err)) return LIME_ERR_IO;
    if (tx.at_rank.size() != n_targ) {
        std::ostringstream m;
        m << "Number of taxIDs = " << tx.at_rank.size() << " lower than genome number: poor taxonomy information to classify.";
        g_cls_err = m.str();
        return LIME_ERR_ARG;
    }
    std::ofstream out(path_out);
    if (!out.is_open()) { g_cls_err = "ERROR: File Output not Open"; return LIME_ERR_IO; }
    out << "C/U/A/H,IdSeqRead,TaxID,maxSim\n";
    int rc = binary ? run<BinSource>(n_files, inputs, n_reads, n_targ, tx, rank, higher != 0, out, counts, g_cls_err)
                    : run<TxtSource>(n_files, inputs, n_reads, n_targ, tx, rank, higher != 0, out, counts, g_cls_err);
    out.close();
    return rc;
}
