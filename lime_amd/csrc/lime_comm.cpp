// lime_comm.cpp -- the one exchange step of the multi-GPU path, through the C ABI: the per-GPU uint8
// score tables are summed modulo 256 over xGMI with RCCL (ncclReduceScatter / ncclAllReduce on
// ncclUint8 with ncclSum; mod-256 addition is associative, so the result is bit-identical to the
// reference's single table, ClusterBWT_DA.cpp:178-184 / 243-248 under OpenMP's range partition
// ClusterLCP.cpp:150-161).  Two forms:
//   * one process per GPU (torchrun / mpirun style): lime_comm_unique_id on rank 0, the caller carries
//     the 128 bytes to the other ranks by whatever channel it has, lime_comm_init on every rank;
//   * one process driving several GPUs (the drop-in ClusterBWT_DA with LIME_GPUS=k): lime_multi_*.
// librccl is loaded on first use (dlopen), so single-GPU users of the library do not depend on it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lime_hip.h"

// the part of rccl.h this file uses (rccl/rccl.h:40-64, 448-470), declared here so that building the
// library does not need the RCCL headers' HIP-version checks
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { nccl_Sum = 0, nccl_Max = 2, nccl_Uint8 = 1, nccl_Uint32 = 3, nccl_Uint64 = 5 };   // rccl.h: ncclRedOp_t / ncclDataType_t
}

#if __has_include(<rccl/rccl.h>)
// the values declared above, checked against the installed header (its own translation unit: lime_rccl_check.cpp)
#endif

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
thread_local std::string g_comm_err;

int cfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_comm_err = buf;
    return code;
}

int load_rccl()
{
    if (g_rccl.h) return LIME_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return cfail(LIME_ERR_HIP, "cannot load librccl: %s", dlerror());
#define SYM(field, name) if (!(*(void **)(&g_rccl.field) = dlsym(h, name))) return cfail(LIME_ERR_HIP, "librccl lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount"); SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return LIME_OK;
}

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != 0) return cfail(LIME_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_));    \
    } while (0)
#define HIP_TRYC(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return cfail(e_ == hipErrorOutOfMemory ? LIME_ERR_NOMEM : LIME_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
}

struct lime_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    // lime_comm_exchange_records: grow-only buffers, kept for the life of the communicator (an exchange is part of every timed
    // step: nothing is allocated or freed inside it once the sizes have settled)
    uint64_t *d_rows = nullptr, *h_rows = nullptr; size_t rows_cap = 0;       // the ranks' gathered rows (+ this rank's own), device and pinned host
    uint32_t *d_rx = nullptr; size_t rx_cap = 0;                             // records received for this rank's bins
    uint64_t *d_big = nullptr, *d_pad = nullptr; size_t big_cap = 0;          // all ranks' long-cluster records, padded to big_cap each; this rank's padded list
    uint64_t *d_st2 = nullptr, *h_st2 = nullptr;                              // second agreement (a rank had to grow a buffer): one word per rank
};

extern "C" const char *lime_comm_error(void) { return g_comm_err.c_str(); }

extern "C" int lime_comm_unique_id(uint8_t id[LIME_COMM_ID_BYTES])
{
    int rc = load_rccl(); if (rc) return rc;
    if (!id) return cfail(LIME_ERR_ARG, "lime_comm_unique_id: id is NULL");
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    static_assert(sizeof u == LIME_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, sizeof u);
    return LIME_OK;
}

extern "C" int lime_comm_init(const uint8_t id[LIME_COMM_ID_BYTES], int rank, int world, lime_comm **out)
{
    int rc = load_rccl(); if (rc) return rc;
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return cfail(LIME_ERR_ARG, "lime_comm_init: bad argument");
    *out = nullptr;
    lime_comm *c = new (std::nothrow) lime_comm();
    if (!c) return cfail(LIME_ERR_NOMEM, "lime_comm_init: out of host memory");
    c->rank = rank; c->world = world;
    HIP_TRYC(hipGetDevice(&c->device));
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != 0) { delete c; return cfail(LIME_ERR_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    *out = c;
    return LIME_OK;
}

// ranks RCCL itself counts in the communicator (ncclCommCount): the bench line carries it next to WORLD_SIZE
extern "C" int lime_comm_count(lime_comm *c, int *ranks)
{
    if (!c || !ranks) return cfail(LIME_ERR_ARG, "lime_comm_count: NULL argument");
    NCCL_TRY(g_rccl.CommCount(c->comm, ranks));
    return LIME_OK;
}

extern "C" void lime_comm_destroy(lime_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(c->d_rows); (void)hipHostFree(c->h_rows); (void)hipFree(c->d_rx); (void)hipFree(c->d_big); (void)hipFree(c->d_pad);
    (void)hipFree(c->d_st2); (void)hipHostFree(c->h_st2);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

// every rank holds a whole table of world * block_bytes bytes (zero padded past n_reads * n_refs); rank r ends
// with the sum modulo 256 of everybody's block r -- its block of read rows, on which it goes on alone
// (clusterChoose is row-independent, ClusterBWT_DA.cpp:385-443).  Asynchronous on `stream`.
extern "C" int lime_comm_reduce_scatter_tables(lime_comm *c, const uint8_t *d_sim, uint8_t *d_block, size_t block_bytes, void *stream)
{
    if (!c || !d_sim || !d_block) return cfail(LIME_ERR_ARG, "lime_comm_reduce_scatter_tables: NULL argument");
    NCCL_TRY(g_rccl.ReduceScatter(d_sim, d_block, block_bytes, nccl_Uint8, nccl_Sum, c->comm, (hipStream_t)stream));
    return LIME_OK;
}

// the whole-table form (every rank ends with the complete table), in place
extern "C" int lime_comm_allreduce_tables(lime_comm *c, uint8_t *d_sim, size_t bytes, void *stream)
{
    if (!c || !d_sim) return cfail(LIME_ERR_ARG, "lime_comm_allreduce_tables: NULL argument");
    NCCL_TRY(g_rccl.AllReduce(d_sim, d_sim, bytes, nccl_Uint8, nccl_Sum, c->comm, (hipStream_t)stream));
    return LIME_OK;
}

// cluster count (sum) and longest cluster (max) over the ranks: d_sum_max = device array {sum operand, max operand}
extern "C" int lime_comm_combine_counters(lime_comm *c, uint64_t *d_sum_max, void *stream)
{
    if (!c || !d_sum_max) return cfail(LIME_ERR_ARG, "lime_comm_combine_counters: NULL argument");
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY(g_rccl.AllReduce(d_sum_max, d_sum_max, 1, nccl_Uint64, nccl_Sum, c->comm, (hipStream_t)stream));
    NCCL_TRY(g_rccl.AllReduce(d_sum_max + 1, d_sum_max + 1, 1, nccl_Uint64, nccl_Max, c->comm, (hipStream_t)stream));
    NCCL_TRY(g_rccl.GroupEnd());
    return LIME_OK;
}


// ---- owner-partitioned exchange of table updates (large tables: include/lime_hip.h, lime_fused_records_dev) ------------
// Rank r owns the bins [r * per, (r+1) * per) of the table (per = ceil(n_bins / world)): it receives, from every rank, the
// slice of records of those bins (ncclSend / ncclRecv inside one group, after an all-gather of the ranks' bin bases) and
// all ranks' long-cluster records (all-gather), and builds bytes [cell_lo, cell_lo + block_bytes) of the table in d_block
// (capacity: per << bin_shift bytes).  About 4 bytes per update cross xGMI instead of T (G-1)/G, and a rank writes T/G
// bytes of table instead of T.
// A COLLECTIVE: every rank enters the same RCCL calls or none.  What a rank finds wrong locally (block too small, records of
// another table shape, a long-cluster list that overflowed) travels as a status word in the row that is gathered anyway, and
// all ranks return the same error together; the row also carries the rank's buffer capacities, so that every rank knows
// whether ANY rank has to grow a buffer -- only then a second one-word agreement follows the allocations.  Once the sizes
// have settled an exchange allocates nothing and synchronises the stream ONCE (the slice sizes have to reach the host).
int lime_internal_records_peek(lime_ctx *c, lime_records_t *out, const uint32_t **d_bigrec_n, uint32_t *bigrec_cap);   // lime_api.cpp: no synchronisation

namespace {
template <typename T> int grow_dev(T *&p, size_t &cap, size_t want)
{
    if (want <= cap) return LIME_OK;
    T *q = nullptr;
    const size_t n = want + want / 4;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return cfail(e == hipErrorOutOfMemory ? LIME_ERR_NOMEM : LIME_ERR_HIP, "hipMalloc of %zu bytes: %s", n * sizeof(T), hipGetErrorString(e));
    (void)hipFree(p);
    p = q; cap = n;
    return LIME_OK;
}
}

extern "C" int lime_comm_exchange_records(lime_comm *c, lime_ctx *ctx, uint32_t n_reads, uint32_t n_refs, uint8_t *d_block,
                                          size_t block_cap, uint64_t *cell_lo, uint64_t *block_bytes, void *stream)
{
    if (!c || !ctx || !d_block || !cell_lo || !block_bytes) return cfail(LIME_ERR_ARG, "lime_comm_exchange_records: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const int W = c->world, me = c->rank;
    uint32_t n_bins = 0, bin_shift = 0;
    // ---- local checks: their outcome is this rank's status word, not a return (the others would wait in the all-gather).
    // The layout is a pure function of the table's shape: it fails on every rank or on none.
    int rc = lime_records_layout(ctx, n_reads, n_refs, &n_bins, &bin_shift);
    if (rc) return cfail(rc, "%s", lime_last_error());
    const size_t sim_bytes = lime_sim_bytes(n_reads, n_refs);
    const uint32_t per = (n_bins + (uint32_t)W - 1u) / (uint32_t)W;
    auto first_bin = [&](int p) { return per * (uint32_t)p < n_bins ? per * (uint32_t)p : n_bins; };
    const uint32_t b0 = first_bin(me), b1 = b0 + per < n_bins ? b0 + per : n_bins, nb = b1 - b0;
    *cell_lo = (uint64_t)b0 << bin_shift;
    const uint64_t hi = ((uint64_t)b1 << bin_shift) < sim_bytes ? ((uint64_t)b1 << bin_shift) : sim_bytes;
    *block_bytes = hi > *cell_lo ? hi - *cell_lo : 0;
    uint64_t status = 0;
    std::string local_err;
    if (*block_bytes > block_cap) { status = (uint64_t)(uint32_t)(-LIME_ERR_ARG); local_err = "block of " + std::to_string(*block_bytes) + " bytes, room for " + std::to_string(block_cap); }
    lime_records_t R; const uint32_t *d_bigrec_n = nullptr; uint32_t bigrec_cap = 0;
    rc = lime_internal_records_peek(ctx, &R, &d_bigrec_n, &bigrec_cap);
    if (rc && !status) { status = (uint64_t)(uint32_t)(-rc); local_err = lime_last_error(); }
    if (!rc && !status && (R.n_bins != n_bins || R.bin_shift != bin_shift)) { status = (uint64_t)(uint32_t)(-LIME_ERR_ARG); local_err = "the ctx holds records of another table shape"; }
    // ---- the row: bin bases [0 .. n_bins], long-cluster records, their list's capacity, status, receive capacities
    const size_t ROW = (size_t)n_bins + 6;
    enum { W_NBIG = 1, W_BIGCAP = 2, W_STATUS = 3, W_RXCAP = 4, W_BIGBUF = 5 };       // offsets behind the n_bins + 1 bases
    if (ROW * ((size_t)W + 1) > c->rows_cap) {                 // (first call, or another table shape; a failure HERE cannot be agreed on: nothing has been gathered yet)
        (void)hipStreamSynchronize(st);
        (void)hipFree(c->d_rows); (void)hipHostFree(c->h_rows); c->d_rows = nullptr; c->h_rows = nullptr; c->rows_cap = 0;
        HIP_TRYC(hipMalloc(&c->d_rows, ROW * ((size_t)W + 1) * sizeof(uint64_t)));
        HIP_TRYC(hipHostMalloc(&c->h_rows, ROW * ((size_t)W + 1) * sizeof(uint64_t)));
        c->rows_cap = ROW * ((size_t)W + 1);
    }
    if (!c->d_st2) { HIP_TRYC(hipMalloc(&c->d_st2, ((size_t)W + 1) * sizeof(uint64_t))); HIP_TRYC(hipHostMalloc(&c->h_st2, ((size_t)W + 1) * sizeof(uint64_t))); }
    uint64_t *d_mine = c->d_rows + ROW * (size_t)W, *h_mine = c->h_rows + ROW * (size_t)W;
    for (size_t i = 0; i < ROW; ++i) h_mine[i] = 0;
    h_mine[n_bins + W_BIGCAP] = bigrec_cap; h_mine[n_bins + W_STATUS] = status; h_mine[n_bins + W_RXCAP] = c->rx_cap; h_mine[n_bins + W_BIGBUF] = c->big_cap;
    HIP_TRYC(hipMemcpyAsync(d_mine, h_mine, ROW * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    if (!status) {                                            // bases and the long-cluster count straight from the device: no host round trip before the gather
        HIP_TRYC(hipMemcpyAsync(d_mine, R.d_binbase, ((size_t)n_bins + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        HIP_TRYC(hipMemcpyAsync(d_mine + n_bins + W_NBIG, d_bigrec_n, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));      // (little-endian low word; the high word is zero)
    }
    NCCL_TRY(g_rccl.AllGather(d_mine, c->d_rows, ROW, nccl_Uint64, c->comm, st));
    HIP_TRYC(hipMemcpyAsync(c->h_rows, c->d_rows, ROW * (size_t)W * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRYC(hipStreamSynchronize(st));                       // the one synchronisation of an exchange
    const uint64_t *all = c->h_rows, *base = all + ROW * (size_t)me;
    // ---- the same verdict on every rank
    for (int s = 0; s < W; ++s) {
        const uint64_t *bs = all + ROW * (size_t)s;
        if (bs[n_bins + W_STATUS])
            return cfail(-(int)(uint32_t)bs[n_bins + W_STATUS], "lime_comm_exchange_records: rank %d: %s", s, s == me ? local_err.c_str() : "failed its local checks");
        if (bs[n_bins + W_NBIG] > bs[n_bins + W_BIGCAP])
            return cfail(LIME_ERR_NOMEM, "lime_comm_exchange_records: rank %d: more update records of long clusters (%llu) than their list holds (%llu)", s,
                         (unsigned long long)bs[n_bins + W_NBIG], (unsigned long long)bs[n_bins + W_BIGCAP]);
    }
    // what I receive: source s's records of my bins, one source after the other; what everybody needs (from the same rows)
    std::vector<uint64_t> srcoff((size_t)W * (nb + 1));
    uint64_t rx_total = 0, big_max = 0;
    for (int s = 0; s < W; ++s) {
        const uint64_t *bs = all + ROW * (size_t)s;
        for (uint32_t b = 0; b <= nb; ++b) srcoff[(size_t)s * (nb + 1) + b] = rx_total + (bs[b0 + b] - bs[b0]);
        rx_total += bs[b1] - bs[b0];
        if (bs[n_bins + W_NBIG] > big_max) big_max = bs[n_bins + W_NBIG];
    }
    bool any_grows = false;
    for (int p = 0; p < W; ++p) {
        const uint32_t pb0 = first_bin(p), pb1 = pb0 + per < n_bins ? pb0 + per : n_bins;
        uint64_t need = 0;
        for (int s = 0; s < W; ++s) { const uint64_t *bs = all + ROW * (size_t)s; need += bs[pb1] - bs[pb0]; }
        const uint64_t *bp = all + ROW * (size_t)p;
        if (need + 16 > bp[n_bins + W_RXCAP] || big_max + 2 > bp[n_bins + W_BIGBUF]) any_grows = true;
    }
    if (any_grows) {                                          // somebody allocates: agree on the outcome before the first send or receive
        int arc = LIME_OK;
        if (rx_total + 16 > c->rx_cap) arc = grow_dev(c->d_rx, c->rx_cap, (size_t)rx_total + 16);
        if (!arc && big_max + 2 > c->big_cap) {
            size_t cap_pad = 0, cap_big = 0;
            (void)hipFree(c->d_pad); (void)hipFree(c->d_big); c->d_pad = nullptr; c->d_big = nullptr; c->big_cap = 0;
            arc = grow_dev(c->d_pad, cap_pad, (size_t)big_max + 2);
            if (!arc) arc = grow_dev(c->d_big, cap_big, (cap_pad) * (size_t)W);
            if (!arc) c->big_cap = cap_pad;
        }
        c->h_st2[W] = arc ? (uint64_t)(uint32_t)(-arc) : 0;
        HIP_TRYC(hipMemcpyAsync(c->d_st2 + W, c->h_st2 + W, sizeof(uint64_t), hipMemcpyHostToDevice, st));
        NCCL_TRY(g_rccl.AllGather(c->d_st2 + W, c->d_st2, 1, nccl_Uint64, c->comm, st));
        HIP_TRYC(hipMemcpyAsync(c->h_st2, c->d_st2, (size_t)W * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRYC(hipStreamSynchronize(st));
        for (int s = 0; s < W; ++s)
            if (c->h_st2[s]) return cfail(-(int)(uint32_t)c->h_st2[s], "lime_comm_exchange_records: rank %d could not allocate its receive buffers%s%s", s,
                                          s == me ? ": " : "", s == me ? g_comm_err.c_str() : "");
    }
    NCCL_TRY(g_rccl.GroupStart());
    for (int p = 0; p < W; ++p) {
        const uint32_t pb0 = first_bin(p), pb1 = pb0 + per < n_bins ? pb0 + per : n_bins;
        const uint64_t n_send = base[pb1] - base[pb0];                         // my records of p's bins
        const uint64_t *bs = all + ROW * (size_t)p;
        const uint64_t n_recv = bs[b1] - bs[b0];                               // p's records of my bins
        if (n_send) NCCL_TRY(g_rccl.Send(R.d_recs + base[pb0], n_send, nccl_Uint32, p, c->comm, st));
        if (n_recv) NCCL_TRY(g_rccl.Recv(c->d_rx + srcoff[(size_t)p * (nb + 1)], n_recv, nccl_Uint32, p, c->comm, st));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    // the long clusters' records of every rank, padded to the longest list (few: clusters beyond the in-window limit)
    uint64_t n_big_all = 0;
    if (big_max) {
        const uint64_t mine = base[n_bins + W_NBIG];
        HIP_TRYC(hipMemsetAsync(c->d_pad, 0, big_max * sizeof(uint64_t), st));   // t == 0: no update
        if (mine) HIP_TRYC(hipMemcpyAsync(c->d_pad, R.d_bigrecs, mine * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        NCCL_TRY(g_rccl.AllGather(c->d_pad, c->d_big, big_max, nccl_Uint64, c->comm, st));
        n_big_all = (uint64_t)W * big_max;
    }
    rc = lime_apply_records_dev(ctx, (uint32_t)W, c->d_rx, srcoff.data(), nb, bin_shift, c->d_big, n_big_all,
                                *cell_lo, *block_bytes, d_block, st);
    if (rc) return cfail(rc, "%s", lime_last_error());
    return LIME_OK;                                           // asynchronous from here on: the block is complete when `stream` gets there
}

// the communicators of one process driving several GPUs: created once per device list (ncclCommInitAll costs hundreds of
// milliseconds), kept for the life of the process
static std::mutex g_multi_mu;
static std::map<std::vector<int>, std::vector<ncclComm_t>> g_multi_comms;
static int multi_comms(int n_dev, const int *devs, std::vector<ncclComm_t> **out)
{
    std::lock_guard<std::mutex> g(g_multi_mu);
    std::vector<int> key(devs, devs + n_dev);
    auto it = g_multi_comms.find(key);
    if (it == g_multi_comms.end()) {
        std::vector<ncclComm_t> comms(n_dev, nullptr);
        NCCL_TRY(g_rccl.CommInitAll(comms.data(), n_dev, devs));
        it = g_multi_comms.emplace(key, std::move(comms)).first;
    }
    *out = &it->second;
    return LIME_OK;
}

int lime_internal_upload(int n_arr, const void *const *src, void *const *dst, const size_t *bytes, hipStream_t st);   // lime_api.cpp: through the pinned staging ring

// all devices of one process: device k ends with block k of the summed tables (internal; lime_api.cpp uses it too)
int lime_internal_reduce_scatter(int n_dev, const int *devs, uint8_t *const *d_sim, uint8_t *const *d_blk, size_t blk)
{
    int rc = load_rccl(); if (rc) return rc;
    std::vector<ncclComm_t> *pc = nullptr;
    if ((rc = multi_comms(n_dev, devs, &pc))) return rc;
    std::vector<ncclComm_t> &comms = *pc;
    rc = LIME_OK;
    if (g_rccl.GroupStart() != 0) rc = cfail(LIME_ERR_HIP, "ncclGroupStart failed");
    for (int k = 0; k < n_dev && !rc; ++k) {
        (void)hipSetDevice(devs[k]);
        ncclResult_t r = g_rccl.ReduceScatter(d_sim[k], d_blk[k], blk, nccl_Uint8, nccl_Sum, comms[k], nullptr);
        if (r != 0) rc = cfail(LIME_ERR_HIP, "ncclReduceScatter: %s", g_rccl.GetErrorString(r));
    }
    if (!rc && g_rccl.GroupEnd() != 0) rc = cfail(LIME_ERR_HIP, "ncclGroupEnd failed");
    for (int k = 0; k < n_dev; ++k) { (void)hipSetDevice(devs[k]); (void)hipDeviceSynchronize(); }
    return rc;
}

// ---- one process, several GPUs ------------------------------------------------------------
// The collection is cut into position ranges (tile-aligned, each with a read-ahead halo of LIME_MAX_CLUSTER +
// LIME_TILE positions: ClusterLCP.cpp:150-161 chunking, :196-202 skip, :246-264 straddle); device k scans range k
// into its own table; one reduce-scatter leaves device k with read-row block k; the blocks are copied back into
// the caller's table.  Host arrays may be pageable.
namespace {
struct DevSet {
    int dev = -1; lime_ctx *ctx = nullptr; hipStream_t st = nullptr;
    uint32_t *lcp = nullptr, *da = nullptr; uint8_t *ebwt = nullptr, *sim = nullptr, *blk = nullptr;
    ~DevSet() {
        if (dev < 0) return;
        (void)hipSetDevice(dev);
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        (void)hipFree(lcp); (void)hipFree(da); (void)hipFree(ebwt); (void)hipFree(sim); (void)hipFree(blk);
        if (ctx) lime_shutdown(ctx);
    }
};
}

extern "C" int lime_fused_multi(int n_dev, const int *devices, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt,
                                uint64_t n, uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint8_t *sim,
                                uint64_t *n_clusters, uint64_t *max_len)
{
    if (n_dev < 1 || !sim || (n && (!lcp || !da))) return cfail(LIME_ERR_ARG, "lime_fused_multi: bad argument");
    if (!n_reads || !n_refs) return cfail(LIME_ERR_ARG, "lime_fused_multi: n_reads and n_refs must be > 0");
    int have = lime_device_count();
    if (n_dev > have) return cfail(LIME_ERR_ARG, "lime_fused_multi: %d devices asked, %d visible", n_dev, have);
    int rc;
    if (n_dev > 1 && (rc = load_rccl())) return rc;
    std::vector<int> devs(n_dev);
    for (int k = 0; k < n_dev; ++k) devs[k] = devices ? devices[k] : k;
    const uint64_t halo = (uint64_t)LIME_MAX_CLUSTER + LIME_TILE;
    const uint64_t tiles = (n + LIME_TILE - 1) / LIME_TILE;
    const size_t sim_bytes = lime_sim_bytes(n_reads, n_refs);
    const size_t blk = (sim_bytes + 16u * (size_t)n_dev - 1) / (16u * (size_t)n_dev) * 16u;      // block of one device, 16-byte aligned
    std::vector<DevSet> ds(n_dev);
    std::vector<uint64_t> lo(n_dev), own(n_dev), avail(n_dev);
    for (int k = 0; k < n_dev; ++k) {
        const uint64_t t0 = tiles * (uint64_t)k / n_dev, t1 = tiles * (uint64_t)(k + 1) / n_dev;
        lo[k] = t0 * LIME_TILE < n ? t0 * LIME_TILE : n;
        const uint64_t hi = (k == n_dev - 1) ? n : (t1 * LIME_TILE < n ? t1 * LIME_TILE : n);
        own[k] = hi - lo[k];
        avail[k] = hi < n ? ((hi + halo < n ? hi + halo : n) - lo[k]) : n - lo[k];
    }
    // a host thread per device: context, buffers, its range through the pinned staging ring (per-thread FILE* readers in the
    // reference, ClusterLCP.cpp:100-123; k devices upload at k times the PCIe rate of one), the pass, its counters
    std::vector<int> rcs(n_dev, LIME_OK);
    std::vector<std::string> errs(n_dev);
    std::vector<lime_stats_t> stats(n_dev);
    {
        std::vector<std::thread> th;
        for (int k = 0; k < n_dev; ++k)
            th.emplace_back([&, k]() {
                DevSet &d = ds[k];
                auto bad = [&](int code, const std::string &what) { rcs[k] = code; errs[k] = what; };
                auto hipok = [&](hipError_t e, const char *what) { if (e == hipSuccess) return true; bad(e == hipErrorOutOfMemory ? LIME_ERR_NOMEM : LIME_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); return false; };
                if (!hipok(hipSetDevice(devs[k]), "hipSetDevice")) return;
                d.dev = devs[k];
                int r = lime_init(devs[k], &d.ctx);
                if (r) { bad(r, std::string("lime_init: ") + lime_last_error()); return; }
                if (!hipok(hipStreamCreateWithFlags(&d.st, hipStreamNonBlocking), "hipStreamCreate")) return;
                if (!hipok(hipMalloc(&d.lcp, avail[k] * 4 + 16), "hipMalloc") || !hipok(hipMalloc(&d.da, avail[k] * 4 + 16), "hipMalloc")) return;
                if (ebwt && !hipok(hipMalloc(&d.ebwt, avail[k] + 16), "hipMalloc")) return;
                if (!hipok(hipMalloc(&d.sim, blk * n_dev), "hipMalloc") || !hipok(hipMalloc(&d.blk, blk), "hipMalloc")) return;
                if (blk * n_dev > sim_bytes && !hipok(hipMemsetAsync(d.sim + sim_bytes, 0, blk * n_dev - sim_bytes, d.st), "hipMemsetAsync")) return;
                const void *src[3] = {lcp + lo[k], da + lo[k], ebwt ? ebwt + lo[k] : nullptr};
                void *dst[3] = {d.lcp, d.da, d.ebwt};
                const size_t bytes[3] = {(size_t)avail[k] * 4, (size_t)avail[k] * 4, ebwt ? (size_t)avail[k] : 0};
                if ((r = lime_internal_upload(ebwt ? 3 : 2, src, dst, bytes, d.st))) { bad(r, std::string("upload: ") + lime_last_error()); return; }
                r = lime_fused_dev(d.ctx, d.lcp, d.da, d.ebwt, own[k], avail[k], lo[k] + avail[k] == n, n_reads, n_refs, alpha, d.sim, 1, d.st);
                if (r) { bad(r, lime_last_error()); return; }
                r = lime_get_stats(d.ctx, &stats[k], d.st);
                if (r && !(r == LIME_ERR_HALO && (stats[k].edge & LIME_EDGE_OPEN))) bad(r, lime_last_error());
            });
        for (auto &t : th) t.join();
    }
    uint64_t tot = 0, mx = 0;
    std::vector<uint32_t> edges(n_dev);
    for (int k = 0; k < n_dev; ++k) {
        if (rcs[k]) return cfail(rcs[k], "device %d: %s", devs[k], errs[k].c_str());
        tot += stats[k].n_clusters; if (stats[k].max_len > mx) mx = stats[k].max_len;
        edges[k] = stats[k].edge;
    }
    // runs longer than the halo across range borders: clusters among them are refused, the others are nothing
    if ((rc = lime_combine_edges(edges.data(), (uint32_t)n_dev))) return cfail(rc, "%s", lime_last_error());
    if (n_clusters) *n_clusters = tot;
    if (max_len) *max_len = mx;
    if (n_dev == 1) {
        HIP_TRYC(hipSetDevice(devs[0]));
        HIP_TRYC(hipMemcpyAsync(sim, ds[0].sim, (size_t)n_reads * n_refs, hipMemcpyDeviceToHost, ds[0].st));
        HIP_TRYC(hipStreamSynchronize(ds[0].st));
        return LIME_OK;
    }
    std::vector<ncclComm_t> *pc = nullptr;
    if ((rc = multi_comms(n_dev, devs.data(), &pc))) return rc;
    std::vector<ncclComm_t> &comms = *pc;
    rc = LIME_OK;
    if (g_rccl.GroupStart() != 0) rc = cfail(LIME_ERR_HIP, "ncclGroupStart failed");
    for (int k = 0; k < n_dev && !rc; ++k) {
        (void)hipSetDevice(devs[k]);
        ncclResult_t r = g_rccl.ReduceScatter(ds[k].sim, ds[k].blk, blk, nccl_Uint8, nccl_Sum, comms[k], ds[k].st);
        if (r != 0) rc = cfail(LIME_ERR_HIP, "ncclReduceScatter: %s", g_rccl.GetErrorString(r));
    }
    if (!rc && g_rccl.GroupEnd() != 0) rc = cfail(LIME_ERR_HIP, "ncclGroupEnd failed");
    const size_t total = (size_t)n_reads * n_refs;
    for (int k = 0; k < n_dev && !rc; ++k) {
        const size_t b0 = blk * (size_t)k;
        if (b0 >= total) break;
        const size_t len = total - b0 < blk ? total - b0 : blk;
        (void)hipSetDevice(devs[k]);
        hipError_t e = hipMemcpyAsync(sim + b0, ds[k].blk, len, hipMemcpyDeviceToHost, ds[k].st);
        if (e != hipSuccess) rc = cfail(LIME_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < n_dev; ++k) { (void)hipSetDevice(devs[k]); (void)hipStreamSynchronize(ds[k].st); }
    return rc;
}
