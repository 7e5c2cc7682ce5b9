/*
 * lime_hip.h -- C ABI of liblime_hip.so: the MI355X (gfx950) implementation of LiME's
 * alpha-cluster detection + read x genome similarity accumulation hot path.
 *
 * The reference (veronicaguerrini/LiME) has no library or FFI surface: its boundary is
 * process + argv + files (src/ClusterLCP.cpp:56-71, src/ClusterBWT_DA.cpp:496-529).  The
 * entry points below are what a binding of that path would call; each names the reference
 * code it replaces.  The drop-in executables `ClusterLCP` and `ClusterBWT_DA`
 * (lime_amd/csrc/cli_*.cpp) are thin argv/file shells over this ABI, and
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain C types; no exceptions cross the ABI; every function returns LIME_OK
 * or a negative code, with text in lime_last_error(); one lime_ctx per device per process;
 * calls on one ctx are not thread-safe.  All files/arrays are little-endian, headerless:
 *   lcp  u32[N]  lcp[i] = LCP(suffix i-1, suffix i), lcp[0] = 0
 *   da   u32[N]  document id; ids < n_reads are reads, the others genomes (id - n_reads)
 *   ebwt u8[N]   symbol preceding suffix i
 *   sim  u8[n_reads * n_refs] row-major, sums modulo 256 (Tools.h:68 dataTypeSim = uchar)
 * There is NO CPU fallback: without a usable HIP device lime_init fails.
 */
#ifndef LIME_HIP_H
#define LIME_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIME_OK            0
#define LIME_ERR_ARG      (-1)  /* bad argument (NULL, misaligned, out-of-range cluster, ...) */
#define LIME_ERR_HIP      (-2)  /* HIP runtime error / no device                              */
#define LIME_ERR_NOMEM    (-3)  /* host or device allocation failed                           */
#define LIME_ERR_MAXLEN   (-4)  /* a cluster is longer than LIME_MAX_CLUSTER (ClusterBWT_DA.cpp:558-562) */
#define LIME_ERR_HALO     (-5)  /* shard: a run owned by this shard does not close inside its halo       */
#define LIME_ERR_DOCID    (-6)  /* a da value >= n_reads + n_refs was met while scoring       */
#define LIME_ERR_IO       (-7)  /* file I/O (CLI helpers)                                     */

#define LIME_MAX_CLUSTER  65536u /* Tools.h:33 sizeMaxBuf */
#define LIME_TILE         4096u  /* positions per workgroup tile; shard cuts must be multiples */

typedef struct lime_ctx lime_ctx;

/* == ElementCluster, Tools.h:85-88; record of fileFasta.<alpha>.clrs */
typedef struct { uint64_t pStart, len; } lime_cluster_t;

/* a non-zero cell of a read's table row: the (idRef, sim) pair clusterChoose collects
 * (ClusterBWT_DA.cpp:390-402; the reference's pair_sim holds sim already divided by norm) */
typedef struct { uint32_t id_ref, sim; } lime_pair_t;

/* Counters of the last scan on a ctx (device-resident until lime_get_stats syncs). */
typedef struct {
    uint64_t n_clusters;   /* ClusterLCP.cpp:136 nClusters                                  */
    uint64_t max_len;      /* ClusterLCP.cpp:136 maxLen                                     */
    uint64_t n_updates;    /* table cells incremented (t > 0), ClusterBWT_DA.cpp:178-184    */
    uint32_t n_cross;      /* clusters that crossed a tile edge (scored by the list kernel)  */
    uint32_t n_big;        /* clusters longer than the in-tile limit (scored by the big kernel) */
    uint32_t flags;        /* LIME_FLAG_* */
    uint32_t wave_records_max; /* binned table updates: most update records one wave produced    */
    uint32_t edge;         /* LIME_EDGE_* of a shard (see lime_fused_dev)                        */
    uint32_t reserved;
} lime_stats_t;

#define LIME_FLAG_MAXLEN 1u
#define LIME_FLAG_HALO   2u
#define LIME_FLAG_DOCID  4u
#define LIME_FLAG_BADCLUSTER 8u
#define LIME_FLAG_OVERFLOW 16u   /* an internal cluster list was too small (cannot happen with the default sizing) */
#define LIME_FLAG_POOL_FULL 32u  /* binned table updates: the record pool was too small; lime_get_stats repeats the pass */
#define LIME_FLAG_CAS_FALLBACK 128u /* the pass wanted the binned update path and found no device memory for its records: it ran by compare-and-swap on the table instead (same result, several times slower); the reason is in lime_last_error().  Set by lime_get_stats; not an error */
#define LIME_FLAG_INTERNAL 64u   /* a device-side invariant did not hold (the scan's window hand-out): the pass is invalid, LIME_ERR_HIP */

/* Edge word of a shard (lime_stats_t.edge): what the host needs to decide about a run that crosses shard borders
 * and is longer than the read-ahead halo.  LEAD_*: the positions before the shard's first cluster head (they belong
 * to a run that started in an earlier shard) hold a read / a genome, and whether the shard has a head at all.
 * OPEN*: a run headed in the shard's owned range is still open at the end of its arrays (and already longer than
 * LIME_MAX_CLUSTER), with what it holds so far.  lime_combine_edges decides. */
#define LIME_EDGE_LEAD_HEAD 1u
#define LIME_EDGE_LEAD_R    2u
#define LIME_EDGE_LEAD_G    4u
#define LIME_EDGE_OPEN      8u
#define LIME_EDGE_OPEN_R   16u
#define LIME_EDGE_OPEN_G   32u

/* ---- lifecycle ------------------------------------------------------------------------ */
/* device < 0: keep the process's current HIP device.  Replaces the reference's
 * omp_set_num_threads set-up (ClusterLCP.cpp:73-84). */
int  lime_init(int device, lime_ctx **out);
void lime_shutdown(lime_ctx *ctx);
const char *lime_last_error(void);
void lime_free(void *p);                 /* frees host buffers returned by this library     */
const char *lime_version(void);
int  lime_device_count(void);            /* HIP devices visible to the process (0 if none) */
int  lime_pick_device(unsigned salt);    /* the device with the most free memory (ties by salt, e.g. the pid): for
                                          * LiME_paired.sh's four concurrent ClusterLCP processes (:44-53) */
/* Tuning and test knobs of a ctx, by name (value as text; "" = back to the library's own choice where that exists).  None changes a result;
 * most pick which kernel variant or update path runs.  Not a stable interface -- lime_amd/csrc/lime_api.cpp:set_option is the list:
 * update_path (cas|bin|auto), bin_levels ("one,two"), pool_density, pool_slack, scan_static_pct, second_level (tiles|sweeps), part_split, no_probe,
 * probe_min, force_p64, p64_test_base, max_blocks, choose_free, apply_wide, sort_nt, part_lines, no_staging, force_staging, detect_chunk,
 * score_chunk, force_rccl, io_threads, dense_min, no_direct, apply_group, debug_stats, debug_alloc, poison_cache.
 * The ENVIRONMENT is read by lime_init only: LIME_IO_THREADS (host staging threads; the reference's `threads` argument, ClusterLCP.cpp:73-84)
 * always, and -- only when LIME_TEST_HOOKS=1 is set -- a LIME_<KNOB> variable per knob above (tests, experiments).  bench.py refuses to run
 * under LIME_TEST_HOOKS. */
int  lime_set_option(lime_ctx *ctx, const char *key, const char *value);
/* Device blocks of 64 MB and more that the contexts of this process have released (lime_shutdown, scratch that was replaced by a larger block)
 * stay with the library for the next context instead of going back to the driver: on this platform a hipMalloc that is served from recycled
 * pages waits while the driver clears them (about 30 GB/s: seconds for a record pool).  At most a quarter of the device's
 * memory is held; a failed allocation releases it.  lime_trim_cache gives everything back now; returns the bytes released.  (Which pages are
 * "recycled" is not in this process's hands: pages that ANY process freed stay uncleared until the driver hands them out again -- see lime_reserve.) */
size_t lime_trim_cache(void);
/* Takes `bytes` of device memory (current device) from the driver NOW, in one block, for the large buffers of every context of this process: record
 * pools, binned records and the other blocks of 64 MB and more are carved from it (first fit, 2 MB granules) before anything is asked of the driver,
 * and go back to it when their context closes.  For a process that knows what it will need (a server; bench.py): where that hipMalloc meets
 * recycled pages it takes its 30 ms per GB here, once, at start-up, and no pass pays for an allocation again.  May be called again (another
 * block).  lime_trim_cache releases the reserved blocks nothing is carved from.  LIME_ERR_NOMEM if the driver refuses. */
int lime_reserve(size_t bytes);

/* ---- host-pointer API (pageable host arrays; the library stages them through HBM) ------ */

/* ClusterLCP main scan, src/ClusterLCP.cpp:140-283 (StartOrRemain :14-32, Close :34-43).
 * *clusters: library-owned host buffer (lime_free), ascending pStart = the reference's
 * 1-thread order; *max_len / *n_clusters as written to the aux .out file (:304-308). */
int lime_detect(lime_ctx *ctx, const uint32_t *lcp, const uint32_t *da, uint64_t n,
                uint32_t n_reads, uint32_t alpha,
                lime_cluster_t **clusters, uint64_t *n_clusters, uint64_t *max_len);

/* lime_detect with the records appended to the file `path` (fileFasta.<alpha>.clrs, ClusterLCP.cpp:229-235) as the chunks complete --
 * no list of the whole collection in host memory.  What the drop-in ClusterLCP calls. */
int lime_detect_to_file(lime_ctx *ctx, const uint32_t *lcp, const uint32_t *da, uint64_t n,
                        uint32_t n_reads, uint32_t alpha, const char *path, uint64_t *n_clusters, uint64_t *max_len);

/* The host-pointer entry points take arrays in pageable memory and stage them through pinned buffers on a few host threads.  When an array IS a
 * mapped file (the drop-in programs map fileFasta.lcp / .da / .ebwt / .clrs), registering the mapping lets those threads fill the pinned buffers with
 * pread() from `fd` instead of touching the mapping page by page (ClusterLCP.cpp:100-123 reads with one FILE* per thread).  The library keeps its
 * own duplicate of the descriptor until lime_unregister_file(base).
 * Contract: `base` maps the file from offset 0 (byte k of the mapping = byte k of the file), `bytes` is at most the file's size, the caller does
 * not write through the mapping (MAP_PRIVATE copies would be ignored: the FILE is read), registrations do not overlap, and the range is
 * unregistered BEFORE it is unmapped -- a stale registration would make a later array at the same address read the old file.  A registration
 * that breaks the checkable parts of this (size, overlap) is refused with LIME_ERR_ARG. */
int  lime_register_file(const void *base, size_t bytes, int fd);
void lime_unregister_file(const void *base);

/* clusterAnalyze, src/ClusterBWT_DA.cpp:256-358 (Update_ref_symb :81-105,
 * Analysis_and_updating :107-190 / :192-252).  ebwt == NULL selects the EBWT=0 build.
 * sim: caller-owned n_reads*n_refs bytes; it is OVERWRITTEN with the table for these
 * clusters (the reference starts from zero, :606-611).  Clusters may come in any order. */
int lime_score(lime_ctx *ctx, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
               const lime_cluster_t *clusters, uint64_t n_clusters,
               uint32_t n_reads, uint32_t n_refs, uint8_t *sim);

/* ClusterLCP + clusterAnalyze in one pass over the arrays (no .clrs materialised): the
 * benchmark path, 9 B/symbol (EBWT=1) or 8 B/symbol (ebwt == NULL). */
int lime_fused(lime_ctx *ctx, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt,
               uint64_t n, uint32_t n_reads, uint32_t n_refs, uint32_t alpha,
               uint8_t *sim, uint64_t *n_clusters, uint64_t *max_len);

/* lime_fused for collections of any size: the arrays stream from host memory (pageable or pinned)
 * through HBM in position-range chunks of `chunk` symbols (0 = 64 Mi; rounded up to LIME_TILE), each
 * with a read-ahead halo of LIME_MAX_CLUSTER + LIME_TILE positions, the copy of one chunk overlapping
 * the scan of the previous one; the table stays in HBM until the end.  Same results as lime_fused. */
int lime_fused_stream(lime_ctx *ctx, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt,
                      uint64_t n, uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint64_t chunk,
                      uint8_t *sim, uint64_t *n_clusters, uint64_t *max_len);

/* clusterChoose row scan, src/ClusterBWT_DA.cpp:385-402: per read the maximum cell and the
 * number of non-zero cells.  Normalisation/formatting stays on the host (:404-441). */
int lime_choose(lime_ctx *ctx, const uint8_t *sim, uint32_t n_reads, uint32_t n_refs,
                uint8_t *row_max, uint32_t *row_nnz);

/* clusterAnalyze + clusterChoose in one call (what the ClusterBWT_DA program does between reading
 * its inputs and writing .res.*, ClusterBWT_DA.cpp:605-443): like lime_score, but the table never
 * leaves the device; outputs as lime_choose_pairs_dev.  sim may be NULL (else it receives the table). */
int lime_score_choose(lime_ctx *ctx, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                      const lime_cluster_t *clusters, uint64_t n_clusters,
                      uint32_t n_reads, uint32_t n_refs, uint32_t norm, float beta,
                      uint8_t *row_max, uint64_t *row_off, lime_pair_t **pairs, uint64_t *n_pairs,
                      uint8_t *sim);

/* ---- device-pointer API (arrays already resident in HBM; asynchronous on `stream`) ----- *
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Device arrays must be
 * 16-byte aligned (hipMalloc is) and d_sim must be allocated with lime_sim_bytes() bytes.   */

size_t lime_sim_bytes(uint32_t n_reads, uint32_t n_refs);  /* n_reads*n_refs rounded up to 16 */

/* One shard of a position-range partition (single GPU: n_own = n_avail = n, eof = 1).
 * The arrays hold positions [0, n_avail) of the shard: the first n_own are owned, the rest
 * is the read-ahead halo (the reference's straddle loop, ClusterLCP.cpp:246-264).  A cluster
 * belongs to the shard that owns its first position.  eof != 0: the arrays end at the true
 * end of the collection, so an open run closes at n_avail (ClusterLCP.cpp:244-245).
 * zero_sim != 0: clear d_sim first.  Counters: lime_get_stats.
 * Asynchronous on `stream`, except the FIRST pass on a ctx over 2^28 symbols or more (option probe_min, floor 2^24): it is preceded by a
 * sampled density probe (1/64 .. 1/256 of the windows, counted only) that synchronises the stream once -- the update path and the record pool
 * are chosen from what it finds, so that the one pass LiME_paired.sh:62-68 runs lands right.  A first pass of fewer symbols runs without the
 * probe, on the binned update path where the table allows it, with a record pool for 0.45 update records per owned symbol: 2 x 4 bytes x
 * 0.45 x 1.25 x 1.35 = about 6 bytes of extra HBM per symbol until a pass has measured the density.
 * Device memory taken on a pass's first call (scratch, record pool) is allocated synchronously: see lime_trim_cache for what that can cost. */
int lime_fused_dev(lime_ctx *ctx, const uint32_t *d_lcp, const uint32_t *d_da,
                   const uint8_t *d_ebwt, uint64_t n_own, uint64_t n_avail, int eof,
                   uint32_t n_reads, uint32_t n_refs, uint32_t alpha,
                   uint8_t *d_sim, int zero_sim, void *stream);

/* ---- owner-partitioned exchange of table updates (several GPUs, large tables) ----
 * lime_fused_records_dev: the same pass as lime_fused_dev, but no table is built: the shard's table updates are left as
 * 32-bit records grouped by table bin (bin b = table bytes [b << bin_shift, (b+1) << bin_shift); record = offset inside
 * the bin | t << bin_shift), the updates of clusters longer than the in-window limit as 64-bit records (cell | t << 40).
 * The layout is a pure function of the table's shape: every rank gets the same (lime_records_layout).  After
 * lime_get_stats (which also repeats a pass whose record pool proved too small), lime_records_get returns the device
 * arrays (valid until the next pass on the ctx) and, optionally, the bin bases on the host.  The owner of bins
 * [b0, b0 + nb) gathers the slices d_recs[binbase[b0] .. binbase[b0 + nb]) of every rank into one buffer and calls
 * lime_apply_records_dev, which builds bytes [b0 << bin_shift, ...) of the table in d_block (every byte written; add
 * the long clusters' records of ALL ranks).  lime_comm_exchange_records does the transport with RCCL.
 * Reference counterpart: the cluster-range split over threads adding into one table, ClusterBWT_DA.cpp:630-670. */
typedef struct {
    uint32_t n_bins, bin_shift;
    const uint32_t *d_recs;            /* device: records grouped by bin */
    const uint64_t *d_binbase;         /* device: n_bins + 1 positions in d_recs */
    const uint64_t *d_bigrecs;         /* device: updates of the long clusters, cell | t << 40 */
    uint64_t n_bigrecs;
} lime_records_t;
int lime_records_layout(lime_ctx *ctx, uint32_t n_reads, uint32_t n_refs, uint32_t *n_bins, uint32_t *bin_shift);
int lime_fused_records_dev(lime_ctx *ctx, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt,
                           uint64_t n_own, uint64_t n_avail, int eof, uint32_t n_reads, uint32_t n_refs,
                           uint32_t alpha, void *stream);
int lime_records_get(lime_ctx *ctx, lime_records_t *out, uint64_t *h_binbase /* n_bins + 1, may be NULL */, void *stream);
int lime_apply_records_dev(lime_ctx *ctx, uint32_t n_src, const uint32_t *d_rx, const uint64_t *h_srcoff /* [n_src][nb + 1] */,
                           uint32_t nb, uint32_t bin_shift, const uint64_t *d_bigrecs, uint64_t n_bigrecs,
                           uint64_t cell_lo, uint64_t block_bytes, uint8_t *d_block, void *stream);

/* Runs longer than the halo that cross shard borders: the reference reads on without limit (ClusterLCP.cpp:246-264)
 * and only refuses CLUSTERS longer than LIME_MAX_CLUSTER (ClusterBWT_DA.cpp:558-562).  A shard that ends inside such a
 * run reports it in lime_stats_t.edge (and lime_get_stats returns LIME_ERR_HALO to callers that do not look);
 * edge[k] = the edge word of shard k, shards in position order, the first one starting at position 0 and the last one
 * run with eof != 0.  Returns LIME_OK if no border-crossing run is a read+genome cluster (nothing to score: the
 * shards' tables are complete), LIME_ERR_MAXLEN if one is (the reference fails on such input too). */
int lime_combine_edges(const uint32_t *edge, uint32_t n_shards);

/* Detection only.  *d_clusters: library-owned DEVICE buffer valid until the next
 * lime_detect_dev/lime_shutdown on this ctx; pStart = pos_base + local position.
 * Synchronises `stream` (the record count sizes the output). */
int lime_detect_dev(lime_ctx *ctx, const uint32_t *d_lcp, const uint32_t *d_da,
                    uint64_t n_own, uint64_t n_avail, int eof, uint64_t pos_base,
                    uint32_t n_reads, uint32_t alpha,
                    const lime_cluster_t **d_clusters, uint64_t *n_clusters, uint64_t *max_len,
                    void *stream);

/* Scoring of a device-resident cluster list (pStart local to d_da/d_ebwt). */
int lime_score_dev(lime_ctx *ctx, const uint32_t *d_da, const uint8_t *d_ebwt, uint64_t n,
                   const lime_cluster_t *d_clusters, uint64_t n_clusters,
                   uint32_t n_reads, uint32_t n_refs, uint8_t *d_sim, int zero_sim, void *stream);

int lime_choose_dev(lime_ctx *ctx, const uint8_t *d_sim, uint32_t n_reads, uint32_t n_refs,
                    uint8_t *d_row_max, uint32_t *d_row_nnz, void *stream);

/* clusterChoose with the table left in HBM (ClusterBWT_DA.cpp:385-423): the row scan AND the
 * (idRef, sim) lists of the reads that pass `float(max)/norm > beta` (:404-406) are made on the
 * device; only row_max (n_reads bytes) and the compact lists come back.  HOST outputs:
 * row_max[n_reads], row_off[n_reads+1] (cells of read r = pairs[row_off[r] .. row_off[r+1]),
 * ascending idRef; empty for a read that does not pass), *pairs library-allocated (lime_free). */
int lime_choose_pairs_dev(lime_ctx *ctx, const uint8_t *d_sim, uint32_t n_reads, uint32_t n_refs,
                          uint32_t norm, float beta, uint8_t *row_max, uint64_t *row_off,
                          lime_pair_t **pairs, uint64_t *n_pairs, void *stream);

/* ClusterLCP scan + clusterAnalyze + clusterChoose (ClusterLCP.cpp:140-283, ClusterBWT_DA.cpp:256-358, :385-423) on device-resident arrays of the
 * whole collection, outputs as lime_choose_pairs_dev.  Where the binned update path serves the pass (tables beyond 64 MB, n_refs >= 256) the
 * table is never written or read: its 64 KB regions are built in LDS and give the rows' maxima / non-zero counts, then the passing rows' lists.
 * Elsewhere the table is built and scanned.  *stats (may be NULL) as lime_get_stats.  Synchronises `stream`. */
int lime_fused_choose_dev(lime_ctx *ctx, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt, uint64_t n,
                          uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint32_t norm, float beta,
                          uint8_t *row_max, uint64_t *row_off, lime_pair_t **pairs, uint64_t *n_pairs,
                          lime_stats_t *stats, void *stream);

/* Synthetic inputs of SURVEY.md section 8(d): element i is a pure function of (seed, i0+i).
 * Any of the three outputs may be NULL.  mode 0 = iid, 1 = block-correlated symbols. */
int lime_synth_dev(lime_ctx *ctx, uint64_t seed, uint64_t i0, uint64_t count,
                   uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint32_t mode,
                   uint32_t *d_lcp, uint32_t *d_da, uint8_t *d_ebwt, void *stream);

/* Waits for `stream`, returns the counters of the last *_dev scan, and maps flags to an
 * error code (LIME_ERR_MAXLEN / _HALO / _DOCID) -- stats are filled either way.  The results of a
 * lime_fused_dev pass are final only once this has returned: a pass on the binned update path whose record
 * pool proved too small is repeated here with a larger one (the caller's arrays must still be in place). */
int lime_get_stats(lime_ctx *ctx, lime_stats_t *out, void *stream);

/* Average device time (ms) of the main scan kernel over the launches since the last call,
 * measured with HIP events on the launch stream; enabled by lime_set_timing(ctx, 1). */
int lime_set_timing(lime_ctx *ctx, int on);
int lime_get_timing(lime_ctx *ctx, double *scan_ms_avg, uint64_t *launches);
/* the same with the parts of a lime_fused_dev pass: ms_avg[0] the scan kernel, [1] the whole pass (table clear or
 * table build included), [2] everything after the scan kernel, [3] everything before it */
int lime_get_timing_ex(lime_ctx *ctx, double ms_avg[4], uint64_t *launches);
/* What the passes on this ctx have cost on the host side so far (LiME_paired.sh:62-68 runs every collection once: a cold pass pays all of it):
 * out[0] ms inside device allocations, [1] ms inside the sampled density probe in front of the ctx's first pass (synchronisation included),
 * [2] probes run, [3] passes repeated by lime_get_stats (record pool too small), [4] passes that fell back to compare-and-swap
 * (LIME_FLAG_CAS_FALLBACK), [5] update records per owned symbol as last measured (-1: nothing measured yet), [6] lime_fused_choose_dev calls
 * served without the table, [7] reserved. */
int lime_get_host_times(lime_ctx *ctx, double out[8]);

/* ---- multi-GPU: the one exchange step of the path (RCCL over xGMI; librccl is loaded on first use) ---- *
 * The reference partitions positions over OpenMP threads (ClusterLCP.cpp:150-161, skip :196-202, straddle
 * :246-264) and adds into ONE shared table with `omp atomic` (ClusterBWT_DA.cpp:178-184, 243-248).  Here every
 * GPU scans one position range (lime_fused_dev with n_own / n_avail / eof) into its own table and the tables are
 * summed modulo 256 -- addition modulo 256 is associative, so the result is bit-identical.
 * One process per GPU: rank 0 calls lime_comm_unique_id, carries the bytes to the others, all call lime_comm_init
 * (the process's current HIP device is the rank's GPU).  Error text: lime_comm_error(). */
#define LIME_COMM_ID_BYTES 128
typedef struct lime_comm lime_comm;
int  lime_comm_unique_id(uint8_t id[LIME_COMM_ID_BYTES]);
int  lime_comm_init(const uint8_t id[LIME_COMM_ID_BYTES], int rank, int world, lime_comm **out);
void lime_comm_destroy(lime_comm *comm);
int  lime_comm_count(lime_comm *comm, int *ranks);   /* ncclCommCount: the ranks RCCL counts in the communicator */
const char *lime_comm_error(void);
/* d_sim: world * block_bytes bytes (the table, zero padded); rank r receives block r of the sum in d_block
 * (ncclReduceScatter, ncclUint8, ncclSum).  Asynchronous on `stream`. */
int  lime_comm_reduce_scatter_tables(lime_comm *comm, const uint8_t *d_sim, uint8_t *d_block, size_t block_bytes, void *stream);
int  lime_comm_allreduce_tables(lime_comm *comm, uint8_t *d_sim, size_t bytes, void *stream);   /* whole table everywhere, in place */
/* d_sum_max: device array of two u64: [0] summed over the ranks (cluster count), [1] maximum (longest cluster) */
int  lime_comm_combine_counters(lime_comm *comm, uint64_t *d_sum_max, void *stream);
/* Owner-partitioned exchange of the update records lime_fused_records_dev left on `ctx` (call lime_get_stats first): rank r
 * ends with bytes [*cell_lo, *cell_lo + *block_bytes) of the finished table in d_block -- the bins r * per .. (r+1) * per,
 * per = ceil(n_bins / world), see lime_records_layout; block_cap >= per << bin_shift.  Collective: every rank calls it. */
int  lime_comm_exchange_records(lime_comm *comm, lime_ctx *ctx, uint32_t n_reads, uint32_t n_refs, uint8_t *d_block,
                                size_t block_cap, uint64_t *cell_lo, uint64_t *block_bytes, void *stream);

/* One process, n_dev GPUs (devices == NULL: 0 .. n_dev-1): lime_fused of host arrays with the collection cut into
 * n_dev position ranges, one reduce-scatter of the tables by read-row blocks, the blocks copied back into `sim`.
 * What the drop-in ClusterBWT_DA-side programs use under LIME_GPUS=k. */
int  lime_fused_multi(int n_dev, const int *devices, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt,
                      uint64_t n, uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint8_t *sim,
                      uint64_t *n_clusters, uint64_t *max_len);

/* lime_score_choose on n_dev GPUs of one process (what the drop-in ClusterBWT_DA does under LIME_GPUS=k): the cluster
 * list is cut by position into n_dev parts, every device scores its part into its own table, one RCCL reduce-scatter
 * by read-row blocks, row scan and list compaction per block.  Outputs as lime_score_choose. */
int  lime_score_choose_multi(int n_dev, const int *devices, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                             const lime_cluster_t *clusters, uint64_t n_clusters, uint32_t n_reads, uint32_t n_refs,
                             uint32_t norm, float beta, uint8_t *row_max, uint64_t *row_off, lime_pair_t **pairs,
                             uint64_t *n_pairs);

/* ---- pure host helpers (no device work; used by the CLIs and by CPU-side tests) -------- */
uint8_t lime_sym_index(uint8_t byte);                              /* ClusterBWT_DA.cpp:455-470 */
uint8_t lime_pair_score(const uint8_t cr[16], const uint8_t cg[16]); /* :129-177, host build of the device routine */

/* Writers of the reference's files (byte-identical formats). */
int lime_write_clrs(const char *path, const lime_cluster_t *clusters, uint64_t n_clusters); /* ClusterLCP.cpp:229-235 */
int lime_write_aux(const char *path, uint32_t n_reads, uint32_t n_refs, uint32_t alpha,
                   uint64_t max_len, uint64_t n_clusters);                                  /* ClusterLCP.cpp:294-310 */
int lime_read_aux(const char *path, uint32_t *n_reads, uint32_t *n_refs, uint32_t *alpha,
                  uint64_t *max_len, uint64_t *n_clusters);                                 /* ClusterBWT_DA.cpp:531-551 */
/* clusterChoose output, ClusterBWT_DA.cpp:361-450.  row_max may be NULL (then recomputed). */
int lime_write_res_txt(const char *path, const uint8_t *sim, const uint8_t *row_max,
                       uint32_t n_reads, uint32_t n_refs, uint32_t norm, float beta);
int lime_write_res_bin(const char *path_bin, const char *path_pos, const uint8_t *sim,
                       const uint8_t *row_max, uint32_t n_reads, uint32_t n_refs,
                       uint32_t norm, float beta);
/* the same files from the compact form of lime_choose_pairs_dev / lime_score_choose */
int lime_write_res_txt_pairs(const char *path, const uint8_t *row_max, const uint64_t *row_off,
                             const lime_pair_t *pairs, uint32_t n_reads, uint32_t norm, float beta);
int lime_write_res_bin_pairs(const char *path_bin, const char *path_pos, const uint8_t *row_max,
                             const uint64_t *row_off, const lime_pair_t *pairs, uint32_t n_reads,
                             uint32_t norm, float beta);

/* ---- read assignment from the .res files (the consumer of the path, src/Classify.cpp) ---- *
 * inputs: n_files (2 single-end, 4 paired-end) .res base names, in the reference's argv order
 * (Classify.cpp:352-360); binary != 0 reads base.bin + base.pos (BIN=1, the Makefile default), else
 * base.txt; rank 0..6 as the reference's taxRank; higher != 0 = the HIGHER=1 build.  Writes the
 * reference's classification file; counts = {classified, not classified, ambiguous, higher rank}.
 * Pure host code.  Error text: lime_classify_error(). */
int lime_classify(uint32_t n_files, const char *const *inputs, int binary, uint32_t n_reads,
                  uint32_t n_targ, const char *path_out, const char *path_tax, int rank, int higher,
                  uint64_t counts[4]);
const char *lime_classify_error(void);

#ifdef __cplusplus
}
#endif
#endif /* LIME_HIP_H */
