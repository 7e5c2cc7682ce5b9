// lime_kernels.h -- declarations shared by the kernels (lime_kernels.hip) and the C-ABI
// implementation (lime_api.cpp).  Not part of the public ABI (include/lime_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lime_hip.h"

namespace lime {

constexpr uint32_t WIN = 1024;               // positions a wave owns per window
constexpr uint32_t PPL = WIN / 64;           // consecutive positions per lane (16)
constexpr uint32_t NW = WIN / 64;            // 64-bit mask words per window
constexpr uint32_t HALO = 16;                // read-ahead positions behind a window (>= SMALL_MAX)
constexpr uint32_t WPOS = WIN + HALO;
constexpr int SCAN_WG = 256;                 // 4 independent waves per workgroup of k_score_list
// k_scan: threads per workgroup and waves per SIMD it is compiled for.  ONE workgroup fills a CU (16 waves without ebwt:
// <= 128 VGPRs and <= 160 KB of LDS together; 12 with ebwt: 168 VGPRs): its waves work independently but draw their
// windows from one LDS counter, which is what keeps a CU's waves -- which the SIMDs do not serve equally -- ending
// together (k_scan).  Waves per SIMD count (measured, configs[2]: 2 -> 2.94 ms, 3 -> 1.90 ms, 4 about the same as 3).
#ifndef LIME_SCAN_WG0
#define LIME_SCAN_WG0 1024
#define LIME_SCAN_WAVES0 4
#endif
#ifndef LIME_SCAN_WG1
#define LIME_SCAN_WG1 768
#define LIME_SCAN_WAVES1 3
#endif
// Round 4: the EBWT=1 scan that emits update records (BIN) also runs 16 waves = 4 per SIMD: it fits 128 VGPRs, and its LDS fits a CU
// with a shorter update queue (ScanLdsT); the compare-and-swap EBWT=1 scan stays at 12 waves: at 16 it fits the LDS too (160 064 bytes) but
// not 128 VGPRs -- 8 spilled to scratch, configs[1] 0.247 against 0.216 ms (tools/r04_libs.sh, round 4).
#ifndef LIME_SCAN_WG1B
#define LIME_SCAN_WG1B 1024
#define LIME_SCAN_WAVES1B 4
#endif
template <int EBWT, int BIN> struct ScanCfg {
    static constexpr int wg = EBWT ? (BIN ? LIME_SCAN_WG1B : LIME_SCAN_WG1) : LIME_SCAN_WG0, waves = EBWT ? (BIN ? LIME_SCAN_WAVES1B : LIME_SCAN_WAVES1) : LIME_SCAN_WAVES0;
};
// waves per workgroup of the scan that emits update records (the only one whose waves the host counts: pool regions, producers)
inline uint32_t scan_waves_per_wg(int ebwt, int mode) { return (uint32_t)((ebwt && mode == 0 ? ScanCfg<1, 1>::wg : ScanCfg<0, 1>::wg) / 64); }
constexpr int WGSZ = 512;                    // workgroup of the helper kernels (big clusters, choose, synth)
constexpr uint32_t SMALL_MAX = 16;           // longest cluster scored inside a window
constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t T_SHIFT = 25;             // update-queue entry: genome | t << T_SHIFT
constexpr uint32_t MAX_REFS = 1u << T_SHIFT; // so n_refs must stay below this
constexpr uint32_t HT_BITS = 17;             // >= 2 x LIME_MAX_CLUSTER slots: the table never fills
constexpr uint32_t HT_SIZE = 1u << HT_BITS;
constexpr uint32_t HT_EMPTY = 0xFFFFFFFFu;
// ---- binned table updates (bin-then-apply): the scan appends (cell, t) records to per-wave regions of a
// pool and counts them per table bin; k_part moves them into their bins, k_part2 (bins wider than a region)
// on into their regions; k_apply builds each 64 KB region of the table in LDS from its records and writes it
// out once
constexpr uint32_t BIN_MAX = 3072;           // bins: one u32 counter each in the 12 KB of LDS the CAS slots use otherwise
#ifndef LIME_REGION_SHIFT
#define LIME_REGION_SHIFT 16
#endif
constexpr uint32_t REGION_SHIFT = LIME_REGION_SHIFT;   // k_apply / k_apply_tiles build 2^REGION_SHIFT bytes of the table per workgroup (<= 16: the second level's records are 16-bit)
static_assert(REGION_SHIFT >= 12 && REGION_SHIFT <= 16, "region offsets are 16-bit records");
constexpr uint32_t BIN_ONE_LEVEL = 1024;     // tables of up to this many regions: one bin per region, no second level
constexpr uint32_t BIN_TWO_LEVEL = 2048;     // larger tables: at most this many bins of 2^k regions each (while k allows); measured best of 256..2048 on configs[2]
constexpr uint32_t BIN_SHIFT_MAX = 25;       // bin-relative cell offset + 7 bits of t must fit 32 bits
constexpr uint32_t CELL_BITS = 40;           // a cell (byte index of the table) has at most this many bits
constexpr uint32_t MAX_SUB = 8;              // binned updates: a pool record is the cell's low 32 bits, its high part the number of the wave's sub-region: tables up to 32 GB
constexpr int APPLY_WG = 512;
constexpr uint32_t BIG_GRID = 32;            // workgroups of k_score_big (each owns a scratch table)
constexpr size_t BIG_SCRATCH_WORDS = (size_t)HT_SIZE + (size_t)HT_SIZE * 16u + 2u * LIME_MAX_CLUSTER;

struct TileSummary { uint32_t first_head, last_head, pre, suf; };   // per window: offsets in window; flags bit0 read, bit1 genome
struct CrossRec { uint64_t start, len; };                          // len == 0: none
struct OpenRec { uint64_t start; uint32_t flags, pad; };           // scoring scan: a window's last head whose run has no head in the read-ahead; flags bit0 read, bit1 genome so far

struct DevStats {                            // same layout as lime_stats_t
    unsigned long long n_clusters, max_len, n_updates;
    uint32_t n_cross, n_big, flags, wave_records_max, edge, n_open;    // n_open (lime_stats_t.reserved): records in ScanArgs::open
};
static_assert(sizeof(DevStats) == sizeof(lime_stats_t), "DevStats must mirror lime_stats_t");

// what the count pass keeps of a window for the emit pass: per 16-position chunk the heads of accepted
// owned clusters, all heads, and where the segment of the chunk's last head ends
struct WinMasks { uint16_t ah[64], h[64]; uint32_t e_suf[64]; };     // per 16-position chunk (lane)

struct ScanArgs {
    const uint32_t *lcp; const uint32_t *da; const uint8_t *ebwt;
    uint64_t n_own, n_avail, pos_base;
    int eof;
    uint32_t n_reads, n_refs, alpha, n_tiles;    // n_tiles: number of WIN-position windows
    uint8_t *sim;
    TileSummary *summ; OpenRec *open;            // detection pass: a summary per window; scoring scan: the noted open segments (the same buffer)
    DevStats *stats;
    lime_cluster_t *small; uint32_t cross_cap;   // tile-crossing clusters <= SMALL_MAX
    lime_cluster_t *big; uint32_t big_cap;       // clusters > SMALL_MAX
    uint32_t *tile_cnt; uint64_t *tile_off; CrossRec *cross; lime_cluster_t *out;   // detect only
    WinMasks *wmask;                             // detect only: count pass -> emit pass
    uint32_t *edge;                              // LIME_EDGE_* word of this shard (default: &stats->edge)
    uint32_t no_direct;                          // binned updates: 1 = through the update queue (k_scan<., 0, 1>) even where the scorers could write the records themselves (option no_direct: comparison runs, tests)
    uint32_t dense_min;                          // k_scan: a window with more accepted clusters than this lists its 2-symbol clusters apart (64)
    uint32_t probe_shift;                        // density probe (lime_api.cpp): only every 2^probe_shift-th chunk of a workgroup's wave count of windows is scanned; 0 = a pass
    uint32_t *dyn; uint32_t n_static, static_pct;           // k_scan: rounds of round-robin window chunks before the chunks come from the counter dyn[0] (dyn[1]: workgroups done; both are left at 0); set by the launch wrapper from static_pct
    uint32_t sub_rb, sub_gb;                     // binned, two sub-regions: cell >= 2^32 <=> read > sub_rb or (read == sub_rb and genome >= sub_gb); one sub-region: sub_rb = ~0
    uint32_t *sticky;                            // passes whose record pool overflowed and that lime_get_stats has not settled yet (never cleared by a pass)
    int ablate;                                  // timing experiments only (LIME_ABLATE_BUILD): 0 = full kernel
    // binned table updates (upd_mode 1; 0 = compare-and-swap on the table)
    int upd_mode;
    uint32_t *pool; uint32_t cap_w, n_sub;       // 32-bit records of wave w, sub-region s (= high part of the cell): pool[(w * n_sub + s) * cap_w ..), at most cap_w each
    uint32_t *wave_cnt;                          // [wave * n_sub + s]: records in that sub-region
    uint32_t prod_waves;                         // waves per workgroup of the scan that made them
    uint32_t *counts;                            // [n_bins][gridDim.x]: records of bin b counted by workgroup p
    uint32_t n_bins, bin_shift;                  // bin of a cell = cell >> bin_shift
    uint64_t *bigrec; uint32_t *bigrec_n; uint32_t bigrec_cap;   // sim == NULL (records for an owner-partitioned exchange): the long clusters' updates, cell | t << CELL_BITS
};

// what k_apply_tiles does with a finished region instead of writing it to the table (clusterChoose without the table: lime_fused_choose_dev)
struct ApplyFin {
    uint32_t n_refs; uint64_t table_bytes;          // n_reads * n_refs (no padding)
    uint32_t *row_max, *row_nnz;                    // mode 1 out: a word per read, zeroed beforehand
    uint32_t *last_nnz;                             // per region: the non-zero cells of its last row segment (mode 1 out, mode 2 in)
    const uint64_t *row_off; lime_pair_t *pairs;    // mode 2: where the passing reads' (idRef, sim) lists go
    const uint64_t *big_off; const uint64_t *bigrecs;   // the long clusters' update records bucketed by region (big_off: n_regions + 1), or NULL
    const uint4 *region_rows;                       // per region (launch_region_rows): first row, offset of the region's first byte in it, row segments, bytes | skip << 31
};

void launch_tile(int ebwt, int mode, const ScanArgs &a, uint32_t max_blocks, hipStream_t st);
uint32_t scan_grid(int ebwt, int mode, int binned, uint32_t n_tiles, uint32_t max_blocks, uint32_t probe_shift = 0);   // workgroups launch_tile will use
// binned updates: after the scan (n_prod = its grid) -- per-bin prefix over the producers and bin totals,
// records into bins, table regions from bins
void launch_bin_rowscan(uint32_t *counts, uint32_t *totals, uint32_t n_bins, uint32_t n_prod, hipStream_t st);
void launch_rowscan_resolve(const ScanArgs &a, uint32_t *counts, uint32_t *totals, uint32_t n_bins, uint32_t n_prod, hipStream_t st);   // launch_bin_rowscan + launch_resolve(0, ..) in one launch
void launch_bin_bases(const uint32_t *totals, uint64_t *binbase /* n_bins + 1 */, uint32_t *tbase /* n_bins + 1 tiles before each bin, or NULL */, uint32_t n_bins, hipStream_t st);
void launch_part(const ScanArgs &a, uint32_t n_prod, const uint64_t *binbase, uint32_t *out, hipStream_t st, bool p64 = false, bool lines_ok = true);   // p64: the pool holds 2^32 records or more (64-bit positions in `out`); lines_ok: k_part_lines may serve layouts it fits
void launch_part2(const uint32_t *recs, const uint64_t *binbase, uint32_t n_bins, uint32_t bin_shift, uint64_t *regbase,
                  uint32_t *out, hipStream_t st);
void launch_apply(uint8_t *sim, size_t sim_bytes, const uint32_t *recs, const uint64_t *regbase, uint32_t bin_shift, hipStream_t st);
void launch_resolve(int mode, const ScanArgs &a, hipStream_t st);
void launch_apply_by_tiles(uint8_t *sim, size_t sim_bytes, const uint32_t *recs, const uint64_t *binbase, uint32_t n_bins, uint32_t bin_shift,
                           uint32_t *tbase, uint16_t *idx, uint16_t *out16, bool many_records, hipStream_t st, bool big_rows = false, bool tbase_ready = false);   // tbase_ready: launch_bin_bases filled it   // many_records: about 2e8 and more (a variant of k_apply_tiles); big_rows: about 1e8 records and more (the 16-bit rows written non-temporally)
// the same in two steps, for clusterChoose without the table: the bins' records sorted into tile rows once (launch_sort_tiles), then
// k_apply_tiles in mode 1 (row maxima / non-zero counts) and, after the host's pass test, mode 2 (the passing rows' pairs) on the same rows
void launch_sort_tiles(const uint32_t *recs, const uint64_t *binbase, uint32_t n_bins, uint32_t bin_shift, uint32_t *tbase, uint16_t *idx, uint16_t *out16,
                       hipStream_t st, bool big_rows, bool tbase_ready = false);
void launch_apply_tiles_fin(int mode, size_t sim_bytes, uint32_t bin_shift, const uint32_t *tbase, const uint16_t *idx, const uint16_t *out16, bool many_records,
                            const ApplyFin &fin, hipStream_t st);
void set_apply_group(uint32_t lg);    // option apply_group: lanes per run of k_apply_tiles = 2^lg (1 .. 6), 0 = by the regions per bin; process-wide
void launch_region_rows(uint32_t n_regions, uint32_t n_refs, uint64_t table_bytes, const uint64_t *row_off /* mode 2; NULL for mode 1 */, void *out /* n_regions x 16 bytes */, hipStream_t st);
// the long clusters' update records (cell | t << CELL_BITS) bucketed by 64 KB table region: cnt / cursor: n_regions words (zeroed here), off: n_regions + 1
void launch_bigrec_buckets(const uint64_t *recs, uint32_t n, uint32_t n_regions, uint32_t *cnt, uint32_t *cursor, uint64_t *off, uint64_t *out, hipStream_t st);
uint64_t tiles_bound(uint64_t n_records, uint32_t n_bins);
uint32_t part_tile();
uint32_t row_stride();
void launch_regroup(const uint32_t *rx, const uint64_t *srcoff, uint32_t n_src, uint32_t nb, const uint64_t *dstbase, uint32_t *dst, hipStream_t st);
void launch_apply_bigrecs(const uint64_t *recs, uint64_t n, uint64_t cell_lo, uint64_t cell_hi, uint8_t *block, hipStream_t st);
void launch_emit(const ScanArgs &a, hipStream_t st);
void launch_scan_tiles(const uint32_t *cnt, uint64_t *off, uint32_t n, unsigned long long *total, hipStream_t st);
void launch_score_list(int ebwt, const ScanArgs &a, const lime_cluster_t *list, uint64_t count, uint32_t blocks, hipStream_t st);
void launch_gather_pairs(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs, const uint64_t *row_off,
                         lime_pair_t *pairs, hipStream_t st);
void launch_score_big(int ebwt, const ScanArgs &a, uint32_t *scratch, hipStream_t st);
void launch_choose(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs, uint8_t *row_max,
                   uint32_t *row_nnz, hipStream_t st);
void launch_synth(uint64_t seed, uint64_t i0, uint64_t count, uint32_t n_reads, uint32_t n_refs,
                  uint32_t alpha, uint32_t mode, uint32_t *lcp, uint32_t *da, uint8_t *ebwt, hipStream_t st);
void launch_preload();      // lime_init: load every kernel, fill the launch wrappers' per-device caches
void launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t st, uint32_t count = 1, size_t pitch = 0);   // count arrays of n words, pitch words apart
void launch_add_u64(uint64_t *p, size_t n, uint64_t v, hipStream_t st);             // p[i] += v (tests: LIME_P64_TEST_BASE)
void launch_zero2(void *a, size_t a_bytes, void *b, size_t b_bytes, hipStream_t st);   // a: a multiple of 4 bytes; b: 16-byte aligned, a multiple of 16 bytes

} // namespace lime
