// lime_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for LiME's hot path:
// alpha-cluster detection over lcp/da (reference: src/ClusterLCP.cpp:140-283) and per-cluster
// read x genome similarity accumulation over ebwt/da (reference: src/ClusterBWT_DA.cpp:256-358).
//
// Design: the unit of work is a POSITION; clusters are segments delimited
// by head(i) := lcp[i] < alpha.  A workgroup streams one 4096-position tile with 16-byte
// coalesced loads, stages da + a flag byte per position in LDS, derives 64-bit head / read /
// genome masks with wave ballots, resolves every segment that lies inside the tile with
// bit-scans on those masks, and scores it from LDS.  Segments that leave the tile are closed
// from per-tile summaries by k_resolve and scored by the list kernel; clusters longer than
// the in-tile limit go to a one-workgroup-per-cluster hash kernel.  Integer/byte work only:
// no MFMA, HBM-bound; the score table is updated with 32-bit CAS on the packed byte cells so
// that every cell is exact modulo 256 like the reference's unsigned char.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lime_device.h"
#include "lime_kernels.h"

namespace lime {

// ---- flag byte kept per staged position -------------------------------------------------
constexpr uint32_t F_SYM = 0x0F;
constexpr uint32_t F_HEAD = 0x10;     // valid until the masks are built ...
constexpr uint32_t F_SINGLE = 0x10;   // ... then: this document occurs once in its cluster
constexpr uint32_t F_READ = 0x20;
constexpr uint32_t F_GEN = 0x40;
constexpr uint32_t F_LEADER = 0x80;   // first occurrence of its document in its cluster

struct TileLds {
    uint32_t da[TILE];                // 16 KiB
    uint8_t fl[TILE];                 //  4 KiB
    uint32_t work[TILE];              // 16 KiB packed read-leader work items
    uint64_t H[NWORDS + 1], R[NWORDS], G[NWORDS], A[NWORDS];
    uint32_t apre[NWORDS];
    uint32_t nwork, cnt, upd;
    unsigned long long maxlen;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// exact "cell += t (mod 256)" on the byte table through a 32-bit CAS on the containing word.
// First attempt assumes the word is still zero (tables are sparse), then retries on the
// value the CAS returned.
__device__ __forceinline__ void sim_add(uint8_t *sim, uint64_t cell, uint32_t t)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(sim + (cell & ~3ull));
    const uint32_t sh = (uint32_t)(cell & 3ull) * 8u;
    uint32_t expect = 0u;
    for (;;) {
        uint32_t b = ((expect >> sh) + t) & 255u;
        uint32_t want = (expect & ~(255u << sh)) | (b << sh);
        uint32_t old = atomicCAS(w, expect, want);
        if (old == expect) break;
        expect = old;
    }
}

__device__ __forceinline__ bool any_in_range(const uint64_t *M, uint32_t s, uint32_t e)
{
    uint32_t ws = s >> 6, we = (e - 1u) >> 6;
    uint64_t ms = ~0ull << (s & 63u), me = ~0ull >> (63u - ((e - 1u) & 63u));
    if (ws == we) return (M[ws] & ms & me) != 0ull;
    if (M[ws] & ms) return true;
    for (uint32_t w = ws + 1u; w < we; ++w) if (M[w]) return true;
    return (M[we] & me) != 0ull;
}

// ---- phase: masks from the staged flag bytes (positions [0, nwords*64)) ---------------
__device__ __forceinline__ void build_masks(TileLds &L, uint32_t nwords)
{
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (uint32_t w = wave; w < nwords; w += WGSZ / 64) {
        uint32_t f = L.fl[w * 64u + lane];
        uint64_t h = __ballot((f & F_HEAD) != 0u);
        uint64_t r = __ballot((f & F_READ) != 0u);
        uint64_t g = __ballot((f & F_GEN) != 0u);
        if (lane == 0) { L.H[w] = h; L.R[w] = r; L.G[w] = g; }
    }
}

// ---- phase A: every position finds its segment, accepted clusters are counted, small
// clusters get leader/single flags and their read leaders become work items ------------
// MODE: 0 score, 1 count only, 2 emit records
template <int EBWT, int MODE>
__device__ __forceinline__ void phase_a(TileLds &L, uint32_t nwords, uint64_t tile_lo,
                                        uint64_t n_own, uint64_t n_avail, int eof,
                                        const ScanArgs &a)
{
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (uint32_t w = wave; w < nwords; w += WGSZ / 64) {
        const uint32_t p = w * 64u + lane;
        const uint64_t H0 = L.H[w];
        // start: latest head <= p
        uint32_t s = NONE32, e = NONE32;
        {
            uint64_t m = H0 & (~0ull >> (63u - lane));
            int wd = (int)w;
            while (m == 0ull && wd > 0) { --wd; m = L.H[wd]; }
            if (m) s = (uint32_t)wd * 64u + 63u - (uint32_t)__clzll((long long)m);
        }
        // end: first head > p
        if (s != NONE32) {
            uint64_t m = (lane == 63u) ? 0ull : (H0 & (~0ull << (lane + 1u)));
            uint32_t wd = w;
            while (m == 0ull && wd + 1u < nwords) { ++wd; m = L.H[wd]; }
            if (m) e = wd * 64u + (uint32_t)__builtin_ctzll(m);
        }
        const bool interior = (s != NONE32) && (e != NONE32) && (tile_lo + s < n_own);
        bool accepted = false;
        uint32_t len = 0;
        if (interior) {
            len = e - s;
            accepted = (len >= 2u) && any_in_range(L.R, s, e) && any_in_range(L.G, s, e);
        }
        const bool is_head = interior && (p == s);
        if (is_head && !eof && (tile_lo + e >= n_avail))         // closed by padding, not by data
            atomicOr(&a.stats->flags, LIME_FLAG_HALO);
        const bool acc_head = is_head && accepted;
        if (acc_head) {
            atomicAdd(&L.cnt, 1u);
            atomicMax(&L.maxlen, (unsigned long long)len);
        }
        if (MODE == 2) {
            uint64_t am = __ballot(acc_head);
            if (lane == 0) L.A[w] = am;
        }
        if (MODE == 0) {
            if (acc_head && len > SMALL_MAX) {                   // too long for the in-tile path
                if (len > LIME_MAX_CLUSTER) atomicOr(&a.stats->flags, LIME_FLAG_MAXLEN);
                else {
                    uint32_t k = atomicAdd(&a.stats->n_big, 1u);
                    if (k < a.big_cap) { a.big[k].pStart = tile_lo + s; a.big[k].len = len; }
                }
            }
            bool work = false;
            uint32_t f = L.fl[p];
            if (accepted && len <= SMALL_MAX) {
                const uint32_t mydoc = L.da[p];
                uint32_t before = 0, total = 0;
                for (uint32_t q = s; q < e; ++q) {
                    uint32_t same = (L.da[q] == mydoc);
                    total += same;
                    before += same & (uint32_t)(q < p);
                }
                f &= ~F_HEAD;
                if (total == 1u) f |= F_SINGLE;
                if (before == 0u) f |= F_LEADER;
                L.fl[p] = (uint8_t)f;
                work = (before == 0u) && (f & F_READ);
            }
            uint64_t wm = __ballot(work);
            if (wm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&L.nwork, (uint32_t)__popcll(wm));
                base = __shfl(base, 0);
                if (work) {
                    uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
                    L.work[base + rank] = p | ((p - s) << 12) | ((len - 1u) << 18);
                }
            }
        }
    }
}

// ---- phase B: one lane per read leader; walks its cluster in LDS and updates the table --
template <int EBWT>
__device__ __forceinline__ void phase_b(TileLds &L, const ScanArgs &a)
{
    const uint32_t nwork = L.nwork;
    uint32_t nupd = 0;
    for (uint32_t k = threadIdx.x; k < nwork; k += WGSZ) {
        const uint32_t it = L.work[k];
        const uint32_t p = it & 0xFFFu, s = p - ((it >> 12) & 63u), e = s + ((it >> 18) & 63u) + 1u;
        const uint32_t fp = L.fl[p], rdoc = L.da[p], rsym = fp & F_SYM;
        const bool rsingle = (fp & F_SINGLE) != 0u;
        uint32_t cr[4] = {0u, 0u, 0u, 0u};
        uint32_t rcount = 1u;
        if (!rsingle) {
            rcount = 0u;
            for (uint32_t q = s; q < e; ++q) {
                uint32_t same = (L.da[q] == rdoc);
                rcount += same;
                if (EBWT) hist_add(cr, L.fl[q] & F_SYM, same);
            }
        } else if (EBWT) {
            hist_add(cr, rsym, 1u);
        }
        const uint64_t row = (uint64_t)rdoc * a.n_refs;
        for (uint32_t q = s; q < e; ++q) {
            const uint32_t fq = L.fl[q];
            if ((fq & (F_GEN | F_LEADER)) != (F_GEN | F_LEADER)) continue;
            const uint32_t gdoc = L.da[q];
            uint32_t t;
            if (rsingle && (fq & F_SINGLE)) {
                t = EBWT ? iupac_match(rsym, fq & F_SYM) : 1u;
            } else {
                uint32_t cg[4] = {0u, 0u, 0u, 0u};
                uint32_t gcount = 0u;
                for (uint32_t x = s; x < e; ++x) {
                    uint32_t same = (L.da[x] == gdoc);
                    gcount += same;
                    if (EBWT) hist_add(cg, L.fl[x] & F_SYM, same);
                }
                // counts <= SMALL_MAX < 255: neither the read wrap nor the genome saturation bites
                t = EBWT ? pair_score(cr, cg) : (rcount < gcount ? rcount : gcount);
            }
            if (t) {
                const uint32_t g = gdoc - a.n_reads;
                if (g < a.n_refs) { sim_add(a.sim, row + g, t); ++nupd; }
                else atomicOr(&a.stats->flags, LIME_FLAG_DOCID);
            }
        }
    }
    if (nupd) atomicAdd(&L.upd, nupd);
}

__device__ __forceinline__ void lds_reset(TileLds &L)
{
    if (threadIdx.x == 0) { L.nwork = 0; L.cnt = 0; L.upd = 0; L.maxlen = 0; }
}

// =========================================================================================
// k_tile: the streaming scan.  One workgroup per 4096-position tile (grid-stride).
// =========================================================================================
template <int EBWT, int MODE>
__global__ __launch_bounds__(WGSZ) void k_tile(ScanArgs a)
{
    __shared__ TileLds L;
    const uint32_t tid = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint64_t tile_lo = (uint64_t)tile * TILE;
        lds_reset(L);
        // ---- load: 4 x (16 B lcp + 16 B da + 4 B ebwt) per lane, fully coalesced --------
#pragma unroll
        for (int k = 0; k < TILE / (WGSZ * 4); ++k) {
            const uint32_t idx = (uint32_t)k * (WGSZ * 4) + tid * 4u;
            const uint64_t g = tile_lo + idx;
            uint32_t lv[4], dv[4], bv = 0u;
            if (g + 4u <= a.n_avail) {
                const uint4 l4 = *reinterpret_cast<const uint4 *>(a.lcp + g);
                const uint4 d4 = *reinterpret_cast<const uint4 *>(a.da + g);
                lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w;
                dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
                if (EBWT) bv = *reinterpret_cast<const uint32_t *>(a.ebwt + g);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = g + j < a.n_avail;
                    lv[j] = ok ? a.lcp[g + j] : 0u;
                    dv[j] = ok ? a.da[g + j] : 0u;
                    if (EBWT && ok) bv |= (uint32_t)a.ebwt[g + j] << (8 * j);
                }
            }
            uint32_t fw = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = g + j < a.n_avail;
                uint32_t f = EBWT ? sym_index((bv >> (8 * j)) & 255u) : 0u;
                if (!ok) f = F_HEAD;                                  // padding closes runs
                else {
                    if (lv[j] < a.alpha) f |= F_HEAD;
                    f |= (dv[j] < a.n_reads) ? F_READ : F_GEN;
                }
                fw |= f << (8 * j);
            }
            *reinterpret_cast<uint4 *>(&L.da[idx]) = make_uint4(dv[0], dv[1], dv[2], dv[3]);
            *reinterpret_cast<uint32_t *>(&L.fl[idx]) = fw;
        }
        __syncthreads();
        build_masks(L, NWORDS);
        __syncthreads();
        phase_a<EBWT, MODE>(L, NWORDS, tile_lo, a.n_own, a.n_avail, a.eof, a);
        __syncthreads();
        if (MODE == 0) { phase_b<EBWT>(L, a); __syncthreads(); }

        // ---- tile summary for segments that leave the tile (wave 0, lane = mask word) ---
        if (tid < 64u) {
            const uint64_t h = L.H[tid], r = L.R[tid], g = L.G[tid];
            const uint64_t hw = __ballot(h != 0ull);
            TileSummary sm;
            sm.first_head = NONE32; sm.last_head = NONE32; sm.pre = 0; sm.suf = 0;
            uint32_t fw = 64u, lw = 0u;
            if (hw) {
                fw = (uint32_t)__builtin_ctzll(hw);
                lw = 63u - (uint32_t)__clzll((long long)hw);
                const uint64_t hf = __shfl(h, (int)fw), hl = __shfl(h, (int)lw);
                sm.first_head = fw * 64u + (uint32_t)__builtin_ctzll(hf);
                sm.last_head = lw * 64u + 63u - (uint32_t)__clzll((long long)hl);
            }
            // prefix [0, first_head) (whole tile when there is no head); suffix [last_head, TILE)
            uint64_t pr = r, pg = g, sr = r, sg = g;
            if (hw) {
                const uint32_t fb = sm.first_head & 63u, lb = sm.last_head & 63u;
                const uint64_t below = fb ? (~0ull >> (64u - fb)) : 0ull;
                const uint64_t from = ~0ull << lb;
                pr = (tid < fw) ? r : (tid == fw ? (r & below) : 0ull);
                pg = (tid < fw) ? g : (tid == fw ? (g & below) : 0ull);
                sr = (tid > lw) ? r : (tid == lw ? (r & from) : 0ull);
                sg = (tid > lw) ? g : (tid == lw ? (g & from) : 0ull);
            } else { sr = 0ull; sg = 0ull; }
            const uint32_t pre = (__ballot(pr != 0ull) ? 1u : 0u) | (__ballot(pg != 0ull) ? 2u : 0u);
            const uint32_t suf = (__ballot(sr != 0ull) ? 1u : 0u) | (__ballot(sg != 0ull) ? 2u : 0u);
            if (tid == 0) {
                sm.pre = pre; sm.suf = suf;
                a.summ[tile] = sm;
                if (MODE == 1) a.tile_cnt[tile] = L.cnt;
                if (MODE != 2 && L.cnt) atomicAdd(&a.stats->n_clusters, (unsigned long long)L.cnt);
                if (MODE != 2 && L.maxlen) atomicMax(&a.stats->max_len, L.maxlen);
                if (MODE == 0 && L.upd) atomicAdd(&a.stats->n_updates, (unsigned long long)L.upd);
            }
        }
        if (MODE == 2) {
            // ordered emission: rank of each accepted head inside the tile
            if (tid < 64u) {
                uint32_t c = (uint32_t)__popcll(L.A[tid]);
                uint32_t x = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)tid >= d) x += y; }
                L.apre[tid] = x - c;
            }
            __syncthreads();
            const uint64_t base = a.tile_off[tile];
            const uint32_t wave = tid >> 6, lane = tid & 63u;
            for (uint32_t w = wave; w < NWORDS; w += WGSZ / 64) {
                const uint64_t am = L.A[w];
                if (!((am >> lane) & 1ull)) continue;
                const uint32_t s = w * 64u + lane;
                // end: first head after s
                uint64_t m = (lane == 63u) ? 0ull : (L.H[w] & (~0ull << (lane + 1u)));
                uint32_t wd = w;
                while (m == 0ull && wd + 1u < NWORDS) { ++wd; m = L.H[wd]; }
                const uint32_t e = wd * 64u + (uint32_t)__builtin_ctzll(m);
                const uint32_t rank = L.apre[w] + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
                lime_cluster_t rec; rec.pStart = a.pos_base + tile_lo + s; rec.len = e - s;
                a.out[base + rank] = rec;
            }
            if (tid == 0) {
                const CrossRec cr = a.cross[tile];
                if (cr.len) {
                    lime_cluster_t rec; rec.pStart = a.pos_base + cr.start; rec.len = cr.len;
                    a.out[base + L.cnt] = rec;
                }
            }
        }
        __syncthreads();
    }
}

// =========================================================================================
// k_resolve: closes the segment that leaves each tile from the summaries of the tiles after
// it (the reference's straddle loop, ClusterLCP.cpp:246-264, and EOF closure :244-245).
// One thread per tile.  MODE 0: push to the small / big score lists; MODE 1: record for emit.
// =========================================================================================
template <int MODE>
__global__ __launch_bounds__(WGSZ) void k_resolve(ScanArgs a)
{
    const uint32_t t = blockIdx.x * WGSZ + threadIdx.x;
    if (t >= a.n_tiles) return;
    if (MODE == 1) { CrossRec z; z.start = 0; z.len = 0; a.cross[t] = z; }
    const TileSummary me = a.summ[t];
    if (me.last_head == NONE32) return;
    const uint64_t s = (uint64_t)t * TILE + me.last_head;
    if (s >= a.n_own) return;                                   // owned by the next shard / padding
    uint32_t fl = me.suf;
    uint64_t e = a.n_avail;
    bool closed_by_data = false;
    for (uint32_t u = t + 1u; u < a.n_tiles; ++u) {
        const TileSummary o = a.summ[u];
        fl |= o.pre;
        if (o.first_head != NONE32) { e = (uint64_t)u * TILE + o.first_head; closed_by_data = true; break; }
    }
    if (e >= a.n_avail) { e = a.n_avail; closed_by_data = false; }
    if (!closed_by_data && !a.eof) { atomicOr(&a.stats->flags, LIME_FLAG_HALO); return; }
    const uint64_t len = e - s;
    if (fl != 3u || len < 2u) return;
    atomicAdd(&a.stats->n_clusters, 1ull);
    atomicMax(&a.stats->max_len, (unsigned long long)len);
    if (MODE == 1) {
        CrossRec c; c.start = s; c.len = len; a.cross[t] = c;
        a.tile_cnt[t] += 1u;
    } else {
        if (len > LIME_MAX_CLUSTER) { atomicOr(&a.stats->flags, LIME_FLAG_MAXLEN); return; }
        if (len > SMALL_MAX) {
            uint32_t k = atomicAdd(&a.stats->n_big, 1u);
            if (k < a.big_cap) { a.big[k].pStart = s; a.big[k].len = len; }
        } else {
            uint32_t k = atomicAdd(&a.stats->n_cross, 1u);
            if (k < a.cross_cap) { a.small[k].pStart = s; a.small[k].len = len; }
        }
    }
}

// =========================================================================================
// k_scan_tiles: exclusive prefix sum of the per-tile record counts (one workgroup).
// =========================================================================================
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t *cnt, uint64_t *off,
                                                     uint32_t n, unsigned long long *total)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + tid;
        uint64_t v = (i < n) ? cnt[i] : 0ull, x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint64_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63u) wsum[wave] = x;
        __syncthreads();
        uint64_t wpre = 0;
        for (uint32_t k = 0; k < wave; ++k) wpre += wsum[k];
        const uint64_t c = carry;
        if (i < n) off[i] = c + wpre + x - v;
        __syncthreads();
        if (tid == 1023u) carry = c + wpre + x;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// =========================================================================================
// k_score_list: scores clusters given as (pStart,len) records.  A workgroup gathers a batch of
// 64 clusters (each <= SMALL_MAX long) side by side into the same LDS layout as a tile and
// runs the same mask / phase A / phase B code; longer clusters are pushed to the big list.
// `count_ptr` (device) or `count` gives the number of records.
// =========================================================================================
template <int EBWT>
__global__ __launch_bounds__(WGSZ) void k_score_list(ScanArgs a, const lime_cluster_t *list,
                                                     const uint32_t *count_ptr, uint64_t count,
                                                     uint32_t cap)
{
    __shared__ TileLds L;
    __shared__ uint32_t c_off[LIST_BATCH + 1];
    __shared__ uint64_t c_ps[LIST_BATCH];
    const uint32_t tid = threadIdx.x;
    uint64_t n_list = count_ptr ? (uint64_t)(*count_ptr < cap ? *count_ptr : cap) : count;
    const uint64_t n_batches = (n_list + LIST_BATCH - 1) / LIST_BATCH;
    for (uint64_t b = blockIdx.x; b < n_batches; b += gridDim.x) {
        lds_reset(L);
        if (tid < 64u) {
            const uint64_t c = b * LIST_BATCH + tid;
            uint64_t ps = 0, len = 0;
            if (c < n_list) { ps = list[c].pStart; len = list[c].len; }
            bool bad = (len > LIME_MAX_CLUSTER) || (ps > a.n_avail) || (len > a.n_avail - ps);
            if (bad) { atomicOr(&a.stats->flags, len > LIME_MAX_CLUSTER ? LIME_FLAG_MAXLEN : LIME_FLAG_BADCLUSTER); len = 0; }
            if (len > SMALL_MAX) {
                uint32_t k = atomicAdd(&a.stats->n_big, 1u);
                if (k < a.big_cap) { a.big[k].pStart = ps; a.big[k].len = len; }
                len = 0;
            }
            if (len < 2u) len = 0;                 // a 0/1-symbol cluster cannot hold a read and a genome
            uint32_t x = (uint32_t)len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)tid >= d) x += y; }
            c_off[tid] = x - (uint32_t)len;
            c_ps[tid] = ps;
            if (tid == 63u) c_off[64] = x;
        }
        __syncthreads();
        const uint32_t total = c_off[LIST_BATCH];
        const uint32_t nwords = (total + 1u + 63u) / 64u;   // +1: sentinel head closing the last cluster
        for (uint32_t i = tid; i < nwords * 64u; i += WGSZ) {
            uint32_t f = F_HEAD, d = 0u;
            if (i < total) {
                uint32_t lo = 0, hi = LIST_BATCH - 1;       // last cluster with c_off <= i
                while (lo < hi) { uint32_t mid = (lo + hi + 1u) >> 1; if (c_off[mid] <= i) lo = mid; else hi = mid - 1u; }
                const uint32_t q = i - c_off[lo];
                const uint64_t g = c_ps[lo] + q;
                d = a.da[g];
                f = EBWT ? sym_index(a.ebwt[g]) : 0u;
                if (q == 0u) f |= F_HEAD;
                f |= (d < a.n_reads) ? F_READ : F_GEN;
            }
            L.da[i] = d; L.fl[i] = (uint8_t)f;
        }
        __syncthreads();
        build_masks(L, nwords);
        __syncthreads();
        // every gathered position is "owned"; counters of this pass are not cluster statistics
        phase_a<EBWT, 0>(L, nwords, 0ull, ~0ull, ~0ull, 1, a);
        __syncthreads();
        phase_b<EBWT>(L, a);
        __syncthreads();
        if (tid == 0 && L.upd) atomicAdd(&a.stats->n_updates, (unsigned long long)L.upd);
        __syncthreads();
    }
}

// =========================================================================================
// k_score_big: one workgroup per long cluster (SMALL_MAX < len <= 65536).  Documents are
// counted in a per-workgroup open-addressing table in global scratch (it stays in L2):
// 16 x u32 counters per document, then every (read, genome) pair is scored by one lane.
// Read counts are reduced mod 256 and genome counts saturated at 255 exactly as the
// reference's unsigned chars do (ClusterBWT_DA.cpp:96-97, :123, :206, :222-223).
// =========================================================================================
__device__ __forceinline__ uint32_t ht_hash(uint32_t doc) { return (doc * 2654435761u) >> (32u - HT_BITS); }

template <int EBWT>
__global__ __launch_bounds__(WGSZ) void k_score_big(ScanArgs a, uint32_t *scratch)
{
    __shared__ uint32_t s_nr, s_ng, s_upd;
    const uint32_t tid = threadIdx.x;
    uint32_t *keys = scratch + (size_t)blockIdx.x * BIG_SCRATCH_WORDS;
    uint32_t *cnt = keys + HT_SIZE;
    uint32_t *rlist = cnt + (size_t)HT_SIZE * 16u;
    uint32_t *glist = rlist + LIME_MAX_CLUSTER;
    const uint32_t n_big = a.stats->n_big < a.big_cap ? a.stats->n_big : a.big_cap;
    for (uint32_t c = blockIdx.x; c < n_big; c += gridDim.x) {
        const uint64_t ps = a.big[c].pStart;
        const uint32_t len = (uint32_t)a.big[c].len;
        if (tid == 0) { s_nr = 0; s_ng = 0; s_upd = 0; }
        __syncthreads();
        // ---- count -------------------------------------------------------------------
        for (uint32_t p = tid; p < len; p += WGSZ) {
            const uint32_t doc = a.da[ps + p];
            const uint32_t sym = EBWT ? sym_index(a.ebwt[ps + p]) : 0u;
            uint32_t h = ht_hash(doc);
            for (;;) {
                uint32_t k = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k == HT_EMPTY) {
                    k = atomicCAS(&keys[h], HT_EMPTY, doc);
                    if (k == HT_EMPTY) {
                        if (doc < a.n_reads) rlist[atomicAdd(&s_nr, 1u)] = h;
                        else glist[atomicAdd(&s_ng, 1u)] = h;
                        break;
                    }
                }
                if (k == doc) break;
                h = (h + 1u) & (HT_SIZE - 1u);
            }
            atomicAdd(&cnt[(size_t)h * 16u + sym], 1u);
        }
        __threadfence();
        __syncthreads();
        // ---- pairs: a wave per read, lanes over genomes ---------------------------------
        const uint32_t nr = s_nr, ng = s_ng;
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        uint32_t nupd = 0;
        for (uint32_t ri = wave; ri < nr; ri += WGSZ / 64) {
            const uint32_t hr = rlist[ri];
            const uint32_t rdoc = keys[hr];
            uint32_t cr[4] = {0u, 0u, 0u, 0u};
            uint32_t rcount = cnt[(size_t)hr * 16u] & 255u;
            if (EBWT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) cr[i >> 2] |= (cnt[(size_t)hr * 16u + i] & 255u) << ((i & 3) * 8);
            }
            const uint64_t row = (uint64_t)rdoc * a.n_refs;
            for (uint32_t gi = lane; gi < ng; gi += 64u) {
                const uint32_t hg = glist[gi];
                const uint32_t gdoc = keys[hg];
                uint32_t t;
                if (EBWT) {
                    uint32_t cg[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        uint32_t v = cnt[(size_t)hg * 16u + i];
                        cg[i >> 2] |= (v > 255u ? 255u : v) << ((i & 3) * 8);
                    }
                    t = pair_score(cr, cg);
                } else {
                    uint32_t v = cnt[(size_t)hg * 16u];
                    v = v > 255u ? 255u : v;
                    t = rcount < v ? rcount : v;
                }
                if (t) {
                    const uint32_t g = gdoc - a.n_reads;
                    if (g < a.n_refs) { sim_add(a.sim, row + g, t); ++nupd; }
                    else atomicOr(&a.stats->flags, LIME_FLAG_DOCID);
                }
            }
        }
        if (nupd) atomicAdd(&s_upd, nupd);
        __syncthreads();
        // ---- restore the table to empty ---------------------------------------------------
        for (uint32_t i = tid; i < nr + ng; i += WGSZ) {
            const uint32_t h = i < nr ? rlist[i] : glist[i - nr];
#pragma unroll
            for (int k = 0; k < 16; ++k) cnt[(size_t)h * 16u + k] = 0u;
            keys[h] = HT_EMPTY;
        }
        __threadfence();
        __syncthreads();
        if (tid == 0 && s_upd) atomicAdd(&a.stats->n_updates, (unsigned long long)s_upd);
    }
}

// =========================================================================================
// k_choose: per read row maximum and non-zero count (clusterChoose row scan,
// ClusterBWT_DA.cpp:385-402).  One wave per row; aligned 32-bit loads over the row's bytes.
// =========================================================================================
__global__ __launch_bounds__(WGSZ) void k_choose(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs,
                                                 uint8_t *row_max, uint32_t *row_nnz)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t waves = (uint64_t)gridDim.x * (WGSZ / 64);
    for (uint64_t r = (uint64_t)blockIdx.x * (WGSZ / 64) + (threadIdx.x >> 6); r < n_reads; r += waves) {
        const uint64_t b0 = r * n_refs, b1 = b0 + n_refs;
        const uint64_t w0 = b0 >> 2, w1 = (b1 + 3ull) >> 2;
        uint32_t mx = 0, nz = 0;
        for (uint64_t w = w0 + lane; w < w1; w += 64u) {
            uint32_t v = reinterpret_cast<const uint32_t *>(sim)[w];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t byte = w * 4ull + k;
                const uint32_t x = (byte >= b0 && byte < b1) ? ((v >> (8 * k)) & 255u) : 0u;
                mx = x > mx ? x : mx;
                nz += (x != 0u);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            uint32_t om = __shfl_xor(mx, d), on = __shfl_xor(nz, d);
            mx = om > mx ? om : mx; nz += on;
        }
        if (lane == 0) { row_max[r] = (uint8_t)mx; row_nnz[r] = nz; }
    }
}

// =========================================================================================
// k_synth: synthetic lcp/da/ebwt, element i a pure function of (seed, i0+i).
// =========================================================================================
__global__ __launch_bounds__(WGSZ) void k_synth(uint64_t seed, uint64_t i0, uint64_t count,
                                                uint32_t n_reads, uint32_t n_refs, uint32_t alpha,
                                                uint32_t mode, uint32_t *lcp, uint32_t *da, uint8_t *ebwt)
{
    const uint64_t stride = (uint64_t)gridDim.x * WGSZ;
    for (uint64_t i = (uint64_t)blockIdx.x * WGSZ + threadIdx.x; i < count; i += stride) {
        uint32_t l, d, s;
        synth_element(seed, i0 + i, n_reads, n_refs, alpha, mode, l, d, s);
        if (lcp) lcp[i] = l;
        if (da) da[i] = d;
        if (ebwt) ebwt[i] = (uint8_t)s;
    }
}

__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// ---- launch wrappers (host) ------------------------------------------------------------
static inline uint32_t tile_grid(uint32_t n_tiles, uint32_t max_blocks)
{
    uint32_t g = n_tiles < max_blocks ? n_tiles : max_blocks;
    return g ? g : 1u;
}

void launch_tile(int ebwt, int mode, const ScanArgs &a, uint32_t max_blocks, hipStream_t st)
{
    const dim3 grid(tile_grid(a.n_tiles, max_blocks)), block(WGSZ);
    if (mode == 0) {
        if (ebwt) hipLaunchKernelGGL((k_tile<1, 0>), grid, block, 0, st, a);
        else      hipLaunchKernelGGL((k_tile<0, 0>), grid, block, 0, st, a);
    } else if (mode == 1) hipLaunchKernelGGL((k_tile<0, 1>), grid, block, 0, st, a);
    else                  hipLaunchKernelGGL((k_tile<0, 2>), grid, block, 0, st, a);
}

void launch_resolve(int mode, const ScanArgs &a, hipStream_t st)
{
    const dim3 grid((a.n_tiles + WGSZ - 1) / WGSZ), block(WGSZ);
    if (mode == 0) hipLaunchKernelGGL((k_resolve<0>), grid, block, 0, st, a);
    else           hipLaunchKernelGGL((k_resolve<1>), grid, block, 0, st, a);
}

void launch_scan_tiles(const uint32_t *cnt, uint64_t *off, uint32_t n, unsigned long long *total, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, cnt, off, n, total);
}

void launch_score_list(int ebwt, const ScanArgs &a, const lime_cluster_t *list, const uint32_t *count_ptr,
                       uint64_t count, uint32_t cap, uint32_t blocks, hipStream_t st)
{
    if (ebwt) hipLaunchKernelGGL((k_score_list<1>), dim3(blocks), dim3(WGSZ), 0, st, a, list, count_ptr, count, cap);
    else      hipLaunchKernelGGL((k_score_list<0>), dim3(blocks), dim3(WGSZ), 0, st, a, list, count_ptr, count, cap);
}

void launch_score_big(int ebwt, const ScanArgs &a, uint32_t *scratch, hipStream_t st)
{
    if (ebwt) hipLaunchKernelGGL((k_score_big<1>), dim3(BIG_GRID), dim3(WGSZ), 0, st, a, scratch);
    else      hipLaunchKernelGGL((k_score_big<0>), dim3(BIG_GRID), dim3(WGSZ), 0, st, a, scratch);
}

void launch_choose(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs, uint8_t *row_max,
                   uint32_t *row_nnz, hipStream_t st)
{
    uint64_t blocks = ((uint64_t)n_reads + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    if (!blocks) blocks = 1;
    hipLaunchKernelGGL(k_choose, dim3((uint32_t)blocks), dim3(WGSZ), 0, st, sim, n_reads, n_refs, row_max, row_nnz);
}

void launch_synth(uint64_t seed, uint64_t i0, uint64_t count, uint32_t n_reads, uint32_t n_refs,
                  uint32_t alpha, uint32_t mode, uint32_t *lcp, uint32_t *da, uint8_t *ebwt, hipStream_t st)
{
    uint64_t blocks = (count + WGSZ - 1) / WGSZ;
    if (blocks > 16384) blocks = 16384;
    if (!blocks) blocks = 1;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)blocks), dim3(WGSZ), 0, st, seed, i0, count, n_reads, n_refs,
                       alpha, mode, lcp, da, ebwt);
}

void launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t st)
{
    hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, st, p, n, v);
}

} // namespace lime
