// lime_host.cpp -- readers/writers of the reference's on-disk formats.
// Pure host code, no device work; shared by the drop-in executables and the Python mirror.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "lime_hip.h"

namespace {
struct File {
    FILE *f;
    File(const char *path, const char *mode) : f(fopen(path, mode)) {}
    ~File() { if (f) fclose(f); }
    int close() { int r = f ? fclose(f) : -1; f = nullptr; return r; }
};
struct PairSim { float sim; uint32_t id; };   // pair_sim, Tools.h:95-98 (8 bytes)
}

// fileFasta.<alpha>.clrs: raw ElementCluster records (ClusterLCP.cpp:229-235)
extern "C" int lime_write_clrs(const char *path, const lime_cluster_t *clusters, uint64_t n)
{
    File o(path, "wb");
    if (!o.f) return LIME_ERR_IO;
    if (n && fwrite(clusters, sizeof(lime_cluster_t), n, o.f) != n) return LIME_ERR_IO;
    return o.close() ? LIME_ERR_IO : LIME_OK;
}

// 28-byte aux file, written field by field (ClusterLCP.cpp:304-308)
extern "C" int lime_write_aux(const char *path, uint32_t n_reads, uint32_t n_refs, uint32_t alpha,
                              uint64_t max_len, uint64_t n_clusters)
{
    File o(path, "wb");
    if (!o.f) return LIME_ERR_IO;
    size_t ok = fwrite(&n_reads, 4, 1, o.f) + fwrite(&n_refs, 4, 1, o.f) + fwrite(&alpha, 4, 1, o.f) +
                fwrite(&max_len, 8, 1, o.f) + fwrite(&n_clusters, 8, 1, o.f);
    if (ok != 5) return LIME_ERR_IO;
    return o.close() ? LIME_ERR_IO : LIME_OK;
}

extern "C" int lime_read_aux(const char *path, uint32_t *n_reads, uint32_t *n_refs, uint32_t *alpha,
                             uint64_t *max_len, uint64_t *n_clusters)
{
    File i(path, "rb");
    if (!i.f) return LIME_ERR_IO;
    size_t ok = fread(n_reads, 4, 1, i.f) + fread(n_refs, 4, 1, i.f) + fread(alpha, 4, 1, i.f) +
                fread(max_len, 8, 1, i.f) + fread(n_clusters, 8, 1, i.f);
    return ok == 5 ? LIME_OK : LIME_ERR_IO;
}

static uint8_t row_maximum(const uint8_t *row, uint32_t n)
{
    uint8_t m = 0;
    for (uint32_t j = 0; j < n; ++j) if (row[j] > m) m = row[j];
    return m;
}

// fileFasta.res.txt (BIN=0): ClusterBWT_DA.cpp:404-441.  A read is written iff
// float(max)/norm > beta (strict, in float); every read terminates its line.
extern "C" int lime_write_res_txt(const char *path, const uint8_t *sim, const uint8_t *row_max,
                                  uint32_t n_reads, uint32_t n_refs, uint32_t norm, float beta)
{
    File o(path, "w");
    if (!o.f) return LIME_ERR_IO;
    std::vector<char> buf(1 << 16);
    setvbuf(o.f, buf.data(), _IOFBF, buf.size());
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t *row = sim + (size_t)r * n_refs;
        const uint8_t m = row_max ? row_max[r] : row_maximum(row, n_refs);
        const float top = static_cast<float>(m) / norm;
        if (top > beta) {
            fprintf(o.f, "%.5f", top);
            for (uint32_t j = 0; j < n_refs; ++j)
                if (row[j]) fprintf(o.f, "\t%u\t%.5f", j, static_cast<float>(row[j]) / norm);
        }
        fputc('\n', o.f);
    }
    return o.close() ? LIME_ERR_IO : LIME_OK;
}

// fileFasta.res.bin + .res.pos (BIN=1): ClusterBWT_DA.cpp:376-436.  .bin = 8-byte records:
// record 0 sentinel {0,0}; per passing read a header {max/norm, count} then count pairs
// {cell/norm, idRef}; .pos = one u64 per read: record index of its header, 0 if none.
extern "C" int lime_write_res_bin(const char *path_bin, const char *path_pos, const uint8_t *sim,
                                  const uint8_t *row_max, uint32_t n_reads, uint32_t n_refs,
                                  uint32_t norm, float beta)
{
    File ob(path_bin, "wb"), op(path_pos, "wb");
    if (!ob.f || !op.f) return LIME_ERR_IO;
    std::vector<PairSim> recs;
    std::vector<uint64_t> pos;
    recs.reserve(1 << 15); pos.reserve(1 << 15);
    uint64_t total = 1;
    PairSim sentinel = {0.0f, 0u};
    if (fwrite(&sentinel, sizeof sentinel, 1, ob.f) != 1) return LIME_ERR_IO;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t *row = sim + (size_t)r * n_refs;
        const uint8_t m = row_max ? row_max[r] : row_maximum(row, n_refs);
        const float top = static_cast<float>(m) / norm;
        if (top > beta) {
            const size_t hdr = recs.size();
            recs.push_back(PairSim{top, 0u});
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < n_refs; ++j)
                if (row[j]) { recs.push_back(PairSim{static_cast<float>(row[j]) / norm, j}); ++cnt; }
            recs[hdr].id = cnt;
            pos.push_back(total);
            total += 1u + cnt;
        } else {
            pos.push_back(0);
        }
        if (recs.size() >= (1u << 15)) {
            if (fwrite(recs.data(), sizeof(PairSim), recs.size(), ob.f) != recs.size()) return LIME_ERR_IO;
            recs.clear();
        }
        if (pos.size() >= (1u << 15)) {
            if (fwrite(pos.data(), 8, pos.size(), op.f) != pos.size()) return LIME_ERR_IO;
            pos.clear();
        }
    }
    if (!recs.empty() && fwrite(recs.data(), sizeof(PairSim), recs.size(), ob.f) != recs.size()) return LIME_ERR_IO;
    if (!pos.empty() && fwrite(pos.data(), 8, pos.size(), op.f) != pos.size()) return LIME_ERR_IO;
    int e1 = ob.close(), e2 = op.close();
    return (e1 || e2) ? LIME_ERR_IO : LIME_OK;
}

// ---- the same two outputs from the compact form (row_max, row_off, pairs) -----------------
extern "C" int lime_write_res_txt_pairs(const char *path, const uint8_t *row_max, const uint64_t *row_off,
                                        const lime_pair_t *pairs, uint32_t n_reads, uint32_t norm, float beta)
{
    File o(path, "w");
    if (!o.f) return LIME_ERR_IO;
    std::vector<char> buf(1 << 16);
    setvbuf(o.f, buf.data(), _IOFBF, buf.size());
    for (uint32_t r = 0; r < n_reads; ++r) {
        const float top = static_cast<float>(row_max[r]) / norm;
        if (top > beta) {
            fprintf(o.f, "%.5f", top);
            for (uint64_t k = row_off[r]; k < row_off[r + 1]; ++k)
                fprintf(o.f, "\t%u\t%.5f", pairs[k].id_ref, static_cast<float>(static_cast<uint8_t>(pairs[k].sim)) / norm);
        }
        fputc('\n', o.f);
    }
    return o.close() ? LIME_ERR_IO : LIME_OK;
}

extern "C" int lime_write_res_bin_pairs(const char *path_bin, const char *path_pos, const uint8_t *row_max,
                                        const uint64_t *row_off, const lime_pair_t *pairs, uint32_t n_reads,
                                        uint32_t norm, float beta)
{
    File ob(path_bin, "wb"), op(path_pos, "wb");
    if (!ob.f || !op.f) return LIME_ERR_IO;
    std::vector<PairSim> recs;
    std::vector<uint64_t> pos;
    recs.reserve(1 << 15); pos.reserve(1 << 15);
    uint64_t total = 1;
    PairSim sentinel = {0.0f, 0u};
    if (fwrite(&sentinel, sizeof sentinel, 1, ob.f) != 1) return LIME_ERR_IO;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const float top = static_cast<float>(row_max[r]) / norm;
        if (top > beta) {
            const uint32_t cnt = (uint32_t)(row_off[r + 1] - row_off[r]);
            recs.push_back(PairSim{top, cnt});
            for (uint64_t k = row_off[r]; k < row_off[r + 1]; ++k)
                recs.push_back(PairSim{static_cast<float>(static_cast<uint8_t>(pairs[k].sim)) / norm, pairs[k].id_ref});
            pos.push_back(total);
            total += 1u + cnt;
        } else {
            pos.push_back(0);
        }
        if (recs.size() >= (1u << 15)) {
            if (fwrite(recs.data(), sizeof(PairSim), recs.size(), ob.f) != recs.size()) return LIME_ERR_IO;
            recs.clear();
        }
        if (pos.size() >= (1u << 15)) {
            if (fwrite(pos.data(), 8, pos.size(), op.f) != pos.size()) return LIME_ERR_IO;
            pos.clear();
        }
    }
    if (!recs.empty() && fwrite(recs.data(), sizeof(PairSim), recs.size(), ob.f) != recs.size()) return LIME_ERR_IO;
    if (!pos.empty() && fwrite(pos.data(), 8, pos.size(), op.f) != pos.size()) return LIME_ERR_IO;
    int e1 = ob.close(), e2 = op.close();
    return (e1 || e2) ? LIME_ERR_IO : LIME_OK;
}
