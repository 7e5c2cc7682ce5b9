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
constexpr uint32_t F_HEAD = 0x10;
constexpr uint32_t F_READ = 0x20;
constexpr uint32_t F_GEN = 0x40;
constexpr uint32_t LDS_PAD = 16;      // clusters are read up to 16 wide from any start

// accepted clusters of a tile, by length class; an entry is start | (len-1) << 12
constexpr uint32_t CAP_A = TILE / 2, CAP_B = TILE / 5 + 1, CAP_C = TILE / 9 + 1, CAP_D = TILE / 2;
constexpr uint32_t QCAP = 128;        // pending table updates per wave
constexpr uint32_t NWAVES = WGSZ / 64;
constexpr uint32_t T_SHIFT = 27;      // queue entry: genome | t << 27 (t <= SMALL_MAX)

struct TileLds {
    uint32_t da[TILE + LDS_PAD];
    uint8_t fl[TILE + LDS_PAD];
    uint16_t listA[CAP_A];            // len 2..4
    uint16_t listB[CAP_B];            // len 5..8
    uint16_t listC[CAP_C];            // len 9..SMALL_MAX
    uint16_t listD[CAP_D];            // clusters with a repeated document (general routine)
    uint32_t q_read[NWAVES][QCAP], q_gen[NWAVES][QCAP];
    uint64_t H[NWORDS], R[NWORDS], G[NWORDS], A[NWORDS];
    uint32_t esuf[NWORDS], apre[NWORDS];
    uint32_t nA, nB, nC, nD, cnt, upd;
    unsigned long long maxlen;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t brev64(uint64_t x) { return __builtin_bitreverse64(x); }
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l)
{
    return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l);
}

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

// ---- masks from the staged flag bytes: wave k owns the 64-bit words [k*WPW, (k+1)*WPW) --
__device__ __forceinline__ void build_masks(TileLds &L)
{
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
#pragma unroll
    for (uint32_t j = 0; j < WPW; ++j) {
        const uint32_t w = wave * WPW + j;
        uint32_t f = L.fl[w * 64u + lane];
        uint64_t h = __ballot((f & F_HEAD) != 0u);
        uint64_t r = __ballot((f & F_READ) != 0u);
        uint64_t g = __ballot((f & F_GEN) != 0u);
        if (lane == 0) { L.H[w] = h; L.R[w] = r; L.G[w] = g; }
    }
}

// ---- phase A, by ONE wave with lane = mask word: all 64 words of the tile at once ---------
// Which heads open an accepted cluster.  Segments wholly inside a word are decided by
// bit-parallel arithmetic on that word's masks; the segment headed at a word's LAST head may
// run into later words: its end and its read/genome content come from the words after it,
// found with bit-scans on the wave-wide "word has a head" ballot.
struct TileCtx {
    uint64_t h, r, g;   // masks of this lane's word
    uint64_t ah;        // heads of accepted, owned clusters that close inside the tile
    uint32_t e_suf;     // tile position where the segment of the word's last head ends (NONE32: leaves the tile)
};

__device__ __forceinline__ TileCtx tile_context(const TileLds &L, uint64_t own_lim)
{
    const uint32_t lane = lane_id();
    TileCtx c;
    c.h = L.H[lane]; c.r = L.R[lane]; c.g = L.G[lane];
    const bool has_h = c.h != 0ull;
    const uint64_t HW = __ballot(has_h), RW = __ballot(c.r != 0ull), GW = __ballot(c.g != 0ull);
    const uint32_t fh = has_h ? (uint32_t)__builtin_ctzll(c.h) : 64u;
    const uint32_t lh = has_h ? 63u - (uint32_t)__clzll((long long)c.h) : 0u;
    const uint64_t lowm = fh >= 64u ? ~0ull : ((1ull << fh) - 1ull);
    const uint32_t pre_r = (c.r & lowm) != 0ull, pre_g = (c.g & lowm) != 0ull;   // headless: whole word
    const uint64_t him = has_h ? (~0ull << lh) : 0ull;
    const uint32_t suf_r = (c.r & him) != 0ull, suf_g = (c.g & him) != 0ull;
    const uint64_t gt = (lane == 63u) ? 0ull : (~0ull << (lane + 1u));
    const uint64_t above = HW & gt;
    const bool has_next = above != 0ull;
    const uint32_t wn = has_next ? (uint32_t)__builtin_ctzll(above) : 64u;
    const uint64_t between = gt & (wn >= 64u ? ~0ull : ((1ull << wn) - 1ull));   // headless words after this one
    const uint32_t mid_r = (RW & between) != 0ull, mid_g = (GW & between) != 0ull;
    const int src = has_next ? (int)wn : (int)lane;
    const uint32_t n_r = __shfl(pre_r, src), n_g = __shfl(pre_g, src), n_fh = __shfl(fh, src);
    c.e_suf = has_next ? wn * 64u + n_fh : NONE32;
    const uint64_t acc_suf = (has_next && (suf_r | mid_r | n_r) && (suf_g | mid_g | n_g)) ? 1ull : 0ull;
    // "segment contains a read / a genome", gathered onto the segment's head bit: in
    // bit-reversed order a head is the TOP of its segment, and adding the seeds to the
    // "may receive from below" mask ripples a carry through each segment up to its head.
    const uint64_t Hr = brev64(c.h), Mr = ~(Hr << 1);
    uint64_t Xr = brev64(c.r), Y = (Xr << 1) & Mr;
    const uint64_t RH = (Xr | (((Mr + Y) ^ Mr) & Mr) | Y) & Hr;
    Xr = brev64(c.g); Y = (Xr << 1) & Mr;
    const uint64_t GH = (Xr | (((Mr + Y) ^ Mr) & Mr) | Y) & Hr;
    uint64_t ah = brev64(RH & GH);
    ah = (ah & ~(1ull << lh)) | (acc_suf << lh);                  // last head: decided with the words after
    const uint64_t wlo = (uint64_t)lane * 64u;                     // ownership (last tiles of a shard)
    ah &= own_lim >= wlo + 64u ? ~0ull : (own_lim <= wlo ? 0ull : ((1ull << (own_lim - wlo)) - 1ull));
    c.ah = has_h ? ah : 0ull;
    return c;
}

// Each lane walks the accepted heads of its word, measures the cluster and files it under its
// length class (MODE 0).  Slots come from wave ballots: no atomics.  Returns via LDS counters.
template <int MODE>
__device__ __forceinline__ void phase_a(TileLds &L, const TileCtx &c, uint64_t tile_lo, const ScanArgs &a)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t ah = c.ah;
    uint32_t nA = 0, nB = 0, nC = 0, maxlen = 0;
    uint32_t n_acc = (uint32_t)__popcll(ah);
    while (__ballot(ah != 0ull)) {
        const bool act = ah != 0ull;
        const uint32_t b = act ? (uint32_t)__builtin_ctzll(ah) : 0u;
        ah &= ah - 1ull;
        const uint32_t p = lane * 64u + b;
        const uint64_t ha = (b == 63u) ? 0ull : (c.h & (~0ull << (b + 1u)));
        const uint32_t e = ha ? lane * 64u + (uint32_t)__builtin_ctzll(ha) : c.e_suf;
        const uint32_t len = act ? e - p : 0u;
        maxlen = len > maxlen ? len : maxlen;
        if (MODE == 0) {
            const bool cA = act && len <= 4u, cB = act && len > 4u && len <= 8u;
            const bool cC = act && len > 8u && len <= SMALL_MAX, cD = act && len > SMALL_MAX;
            const uint16_t item = (uint16_t)(p | ((len - 1u) << 12));
            const uint64_t mA = __ballot(cA), mB = __ballot(cB), mC = __ballot(cC);
            if (cA) L.listA[nA + (uint32_t)__popcll(mA & lt)] = item;
            nA += (uint32_t)__popcll(mA);
            if (mB) { if (cB) L.listB[nB + (uint32_t)__popcll(mB & lt)] = item; nB += (uint32_t)__popcll(mB); }
            if (mC) { if (cC) L.listC[nC + (uint32_t)__popcll(mC & lt)] = item; nC += (uint32_t)__popcll(mC); }
            if (__ballot(cD)) {                                   // rare: too long for the in-tile path
                if (cD) {
                    if (len > LIME_MAX_CLUSTER) atomicOr(&a.stats->flags, LIME_FLAG_MAXLEN);
                    else {
                        uint32_t k = atomicAdd(&a.stats->n_big, 1u);
                        if (k < a.big_cap) { a.big[k].pStart = tile_lo + p; a.big[k].len = len; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t om = __shfl_xor(maxlen, d), on = __shfl_xor(n_acc, d);
        maxlen = om > maxlen ? om : maxlen; n_acc += on;
    }
    if (lane == 0) { L.nA = nA; L.nB = nB; L.nC = nC; L.cnt = n_acc; L.maxlen = maxlen; }
    if (MODE == 2) { L.A[lane] = c.ah; L.esuf[lane] = c.e_suf; }
}

// ---- phase B -------------------------------------------------------------------------------
// A wave's table updates are queued in its own LDS ring and applied together afterwards, so
// that the round trips of the compare-and-swaps overlap instead of following one another.
// (Kept small on purpose: everything below is rolled loops with ONE emit and ONE drain site per
// routine -- an unrolled version of this kernel overflowed the instruction cache.)
struct UpdQueue { uint32_t *qr, *qg; uint32_t n; };

__device__ __forceinline__ void drain(UpdQueue &q, const ScanArgs &a)
{
    if (a.ablate != 5)
        for (uint32_t k = lane_id(); k < q.n; k += 64u) {
            const uint32_t gt = q.qg[k];
            sim_add(a.sim, (uint64_t)q.qr[k] * a.n_refs + (gt & ((1u << T_SHIFT) - 1u)), gt >> T_SHIFT);
        }
    q.n = 0;
}

// every active lane may add one update; all 64 lanes must call (wave ballots inside)
__device__ __forceinline__ uint32_t emit(UpdQueue &q, const ScanArgs &a, bool on, uint32_t rdoc, uint32_t gdoc, uint32_t t)
{
    if (q.n > QCAP - 64u) drain(q, a);
    const uint32_t g = gdoc - a.n_reads;
    const bool bad = on && g >= a.n_refs;
    if (__ballot(bad)) { if (bad) atomicOr(&a.stats->flags, LIME_FLAG_DOCID); }
    on = on && !bad;
    const uint64_t m = __ballot(on);
    if (on) {
        const uint32_t slot = q.n + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
        q.qr[slot] = rdoc; q.qg[slot] = g | (t << T_SHIFT);
    }
    q.n += (uint32_t)__popcll(m);
    return on ? 1u : 0u;
}

// G lanes per cluster (len <= G), element i of the cluster on sub-lane i.  Rotating the group
// shows every lane every other element: a repeated document sends the cluster to the general
// list; otherwise each read lane scores each genome it meets (distinct documents: the pair
// score is iupac_match of the two symbols with EBWT, 1 without).
template <int EBWT, int G>
__device__ __forceinline__ uint32_t cluster_group(TileLds &L, UpdQueue &qu, const ScanArgs &a,
                                                  const uint16_t *list, uint32_t n, uint32_t k_base)
{
    const uint32_t lane = lane_id(), sub = lane & (G - 1u), gbase = lane & ~(G - 1u);
    const uint32_t k = k_base + lane / G;
    const bool on = k < n;
    const uint32_t item = list[on ? k : 0u];
    const uint32_t s = item & 0xFFFu, len = on ? (item >> 12) + 1u : 0u;
    const bool have = sub < len;
    const uint32_t d = L.da[s + sub];
    const uint32_t f = have ? L.fl[s + sub] : 0u;
    uint32_t dup = 0, hits = 0;
    for (uint32_t r = 1; r < (uint32_t)G; ++r) {
        const int partner = (int)(gbase | ((sub + r) & (G - 1u)));
        const uint32_t pd = __shfl(d, partner), pf = __shfl(f, partner);
        const bool both = have && (pf & (F_READ | F_GEN));
        dup |= (uint32_t)(both && pd == d);
        const bool pair = both && (f & F_READ) && (pf & F_GEN);
        const uint32_t t = EBWT ? iupac_match(f & F_SYM, pf & F_SYM) : 1u;
        hits |= (uint32_t)(pair && t) << r;
    }
    const uint64_t dm = __ballot(dup != 0u);
    const bool gdup = ((dm >> gbase) & ((G >= 64) ? ~0ull : ((1ull << G) - 1ull))) != 0ull;
    if (gdup && sub == 0u) L.listD[atomicAdd(&L.nD, 1u)] = (uint16_t)item;
    if (gdup) hits = 0u;
    uint32_t nupd = 0;
    for (uint32_t r = 1; r < (uint32_t)G; ++r) {
        const bool hit = (hits >> r) & 1u;
        if (__ballot(hit) == 0ull) continue;
        const int partner = (int)(gbase | ((sub + r) & (G - 1u)));
        const uint32_t pd = __shfl(d, partner);
        nupd += emit(qu, a, hit, d, pd, 1u);
    }
    return nupd;
}

// General routine, one lane per cluster [s, s+len) staged in LDS (len <= SMALL_MAX), any mix of
// repeated documents.  For every read (first occurrence) and every genome (first occurrence)
// the counts / 16-bin histograms are rebuilt by walking the cluster.  Quadratic, rare.
template <int EBWT>
__device__ __forceinline__ uint32_t cluster_general(TileLds &L, UpdQueue &qu, const ScanArgs &a, bool on, uint32_t s, uint32_t len)
{
    const uint32_t e = on ? s + len : s;
    uint32_t nupd = 0;
    for (uint32_t p = s; __ballot(p < e); ++p) {
        const bool pon = p < e;
        const uint32_t fp = pon ? L.fl[p] : 0u;
        const bool isr = pon && (fp & F_READ);
        const uint32_t rdoc = isr ? L.da[p] : 0u;
        uint32_t earlier = 0, rcount = 0;
        uint32_t cr[4] = {0u, 0u, 0u, 0u};
        if (isr)
            for (uint32_t q = s; q < e; ++q) {
                const uint32_t same = (L.da[q] == rdoc);
                rcount += same;
                earlier |= same & (uint32_t)(q < p);
                if (EBWT) hist_add(cr, L.fl[q] & F_SYM, same);
            }
        const bool rlead = isr && !earlier;
        for (uint32_t q = s; __ballot(rlead && q < e); ++q) {
            const bool qon = rlead && q < e;
            const uint32_t fq = qon ? L.fl[q] : 0u;
            const bool isg = qon && (fq & F_GEN);
            const uint32_t gdoc = isg ? L.da[q] : 0u;
            uint32_t gearlier = 0, gcount = 0;
            uint32_t cg[4] = {0u, 0u, 0u, 0u};
            if (isg)
                for (uint32_t x = s; x < e; ++x) {
                    const uint32_t same = (L.da[x] == gdoc);
                    gcount += same;
                    gearlier |= same & (uint32_t)(x < q);
                    if (EBWT) hist_add(cg, L.fl[x] & F_SYM, same);
                }
            uint32_t t = 0;
            if (isg && !gearlier) t = EBWT ? pair_score(cr, cg) : (rcount < gcount ? rcount : gcount);
            nupd += emit(qu, a, t != 0u, rdoc, gdoc, t);          // counts <= SMALL_MAX: no wrap, no saturation
        }
    }
    return nupd;
}

template <int EBWT>
__device__ __forceinline__ void phase_b(TileLds &L, const ScanArgs &a)
{
    const uint32_t wave = threadIdx.x >> 6;
    UpdQueue qu; qu.qr = L.q_read[wave]; qu.qg = L.q_gen[wave]; qu.n = 0;
    uint32_t nupd = 0;
    const uint32_t nA = L.nA, nB = L.nB, nC = L.nC;
    for (uint32_t k0 = wave * 16u; k0 < nA; k0 += NWAVES * 16u) nupd += cluster_group<EBWT, 4>(L, qu, a, L.listA, nA, k0);
    for (uint32_t k0 = wave * 8u; k0 < nB; k0 += NWAVES * 8u) nupd += cluster_group<EBWT, 8>(L, qu, a, L.listB, nB, k0);
    for (uint32_t k0 = wave * 4u; k0 < nC; k0 += NWAVES * 4u) nupd += cluster_group<EBWT, 16>(L, qu, a, L.listC, nC, k0);
    __syncthreads();                              // clusters with repeated documents are now in list D
    const uint32_t nD = L.nD;
    for (uint32_t k0 = wave * 64u; k0 < nD; k0 += WGSZ) {
        const uint32_t k = k0 + lane_id();
        const uint32_t item = L.listD[k < nD ? k : 0u];
        nupd += cluster_general<EBWT>(L, qu, a, k < nD, item & 0xFFFu, (item >> 12) + 1u);
    }
    drain(qu, a);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) nupd += __shfl_xor(nupd, d);
    if (lane_id() == 0 && nupd) atomicAdd(&L.upd, nupd);
}

__device__ __forceinline__ void lds_reset(TileLds &L)
{
    if (threadIdx.x == 0) { L.nA = 0; L.nB = 0; L.nC = 0; L.nD = 0; L.cnt = 0; L.upd = 0; L.maxlen = 0; }
}

// ---- tile loads: 2 x (16 B lcp + 16 B da + 4 B ebwt) per lane, fully coalesced; kept in
// registers so that the NEXT tile's loads are in flight while the current tile is processed
constexpr int LOAD_K = TILE / (WGSZ * 4);
struct TileRegs { uint32_t lv[LOAD_K][4], dv[LOAD_K][4], bv[LOAD_K]; };

template <int EBWT>
__device__ __forceinline__ void tile_load(TileRegs &t, const ScanArgs &a, uint64_t tile_lo)
{
#pragma unroll
    for (int k = 0; k < LOAD_K; ++k) {
        const uint64_t g = tile_lo + (uint32_t)k * (WGSZ * 4) + threadIdx.x * 4u;
        t.bv[k] = 0u;
        if (g + 4u <= a.n_avail) {
            const uint4 l4 = *reinterpret_cast<const uint4 *>(a.lcp + g);
            const uint4 d4 = *reinterpret_cast<const uint4 *>(a.da + g);
            t.lv[k][0] = l4.x; t.lv[k][1] = l4.y; t.lv[k][2] = l4.z; t.lv[k][3] = l4.w;
            t.dv[k][0] = d4.x; t.dv[k][1] = d4.y; t.dv[k][2] = d4.z; t.dv[k][3] = d4.w;
            if (EBWT) t.bv[k] = *reinterpret_cast<const uint32_t *>(a.ebwt + g);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = g + j < a.n_avail;
                t.lv[k][j] = ok ? a.lcp[g + j] : 0u;
                t.dv[k][j] = ok ? a.da[g + j] : 0u;
                if (EBWT && ok) t.bv[k] |= (uint32_t)a.ebwt[g + j] << (8 * j);
            }
        }
    }
}

template <int EBWT>
__device__ __forceinline__ void tile_stage(TileLds &L, const TileRegs &t, const ScanArgs &a, uint64_t tile_lo)
{
#pragma unroll
    for (int k = 0; k < LOAD_K; ++k) {
        const uint32_t idx = (uint32_t)k * (WGSZ * 4) + threadIdx.x * 4u;
        const uint64_t g = tile_lo + idx;
        uint32_t fw = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t f = EBWT ? sym_index((t.bv[k] >> (8 * j)) & 255u) : 0u;
            if (g + j >= a.n_avail) f = F_HEAD;                       // padding closes runs
            else {
                if (t.lv[k][j] < a.alpha) f |= F_HEAD;
                f |= (t.dv[k][j] < a.n_reads) ? F_READ : F_GEN;
            }
            fw |= f << (8 * j);
        }
        *reinterpret_cast<uint4 *>(&L.da[idx]) = make_uint4(t.dv[k][0], t.dv[k][1], t.dv[k][2], t.dv[k][3]);
        *reinterpret_cast<uint32_t *>(&L.fl[idx]) = fw;
    }
}

// =========================================================================================
// k_tile: the streaming scan.  Persistent 512-thread workgroups walk the 4096-position tiles
// with stride gridDim.x; the next tile's loads are issued before the current one is processed.
// =========================================================================================
template <int EBWT, int MODE>
__global__ __launch_bounds__(WGSZ, 4) void k_tile(ScanArgs a)
{
    __shared__ TileLds L;
    const uint32_t tid = threadIdx.x;
    uint32_t tile = blockIdx.x;
    if (tile >= a.n_tiles) return;
    TileRegs regs;
    tile_load<EBWT>(regs, a, (uint64_t)tile * TILE);
    for (;;) {
        const uint64_t tile_lo = (uint64_t)tile * TILE;
        const uint64_t own_lim = a.n_own > tile_lo ? a.n_own - tile_lo : 0ull;
        lds_reset(L);
        tile_stage<EBWT>(L, regs, a, tile_lo);
        const uint32_t next = tile + gridDim.x;
        if (next < a.n_tiles && a.ablate != 8) tile_load<EBWT>(regs, a, (uint64_t)next * TILE);
        __syncthreads();
        if (a.ablate != 1) {
        build_masks(L);
        __syncthreads();
        if (tid < 64u) {
            // ---- phase A and the tile summary: wave 0, lane = mask word ---------------------
            const TileCtx c = tile_context(L, own_lim);
            if (a.ablate != 3) phase_a<MODE>(L, c, tile_lo, a);
            const uint64_t hw = __ballot(c.h != 0ull);
            TileSummary sm;
            sm.first_head = NONE32; sm.last_head = NONE32; sm.pre = 0; sm.suf = 0;
            uint32_t fw = 64u, lw = 0u;
            if (hw) {
                fw = (uint32_t)__builtin_ctzll(hw);
                lw = 63u - (uint32_t)__clzll((long long)hw);
                const uint64_t hf = rl64(c.h, fw), hl = rl64(c.h, lw);
                sm.first_head = fw * 64u + (uint32_t)__builtin_ctzll(hf);
                sm.last_head = lw * 64u + 63u - (uint32_t)__clzll((long long)hl);
            }
            // prefix [0, first_head) (whole tile when there is no head); suffix [last_head, TILE)
            uint64_t pr = c.r, pg = c.g, sr = 0ull, sg = 0ull;
            if (hw) {
                const uint32_t fb = sm.first_head & 63u, lb = sm.last_head & 63u;
                const uint64_t below = fb ? (~0ull >> (64u - fb)) : 0ull;
                const uint64_t from = ~0ull << lb;
                pr = (tid < fw) ? c.r : (tid == fw ? (c.r & below) : 0ull);
                pg = (tid < fw) ? c.g : (tid == fw ? (c.g & below) : 0ull);
                sr = (tid > lw) ? c.r : (tid == lw ? (c.r & from) : 0ull);
                sg = (tid > lw) ? c.g : (tid == lw ? (c.g & from) : 0ull);
            }
            const uint32_t pre = (__ballot(pr != 0ull) ? 1u : 0u) | (__ballot(pg != 0ull) ? 2u : 0u);
            const uint32_t suf = (__ballot(sr != 0ull) ? 1u : 0u) | (__ballot(sg != 0ull) ? 2u : 0u);
            if (tid == 0) { sm.pre = pre; sm.suf = suf; a.summ[tile] = sm; }
            // a run closed by padding instead of data while more data exists beyond the halo:
            // the last data head of the tile is owned and nothing but padding follows it
            if (!a.eof && tile_lo + TILE > a.n_avail) {
                const uint64_t lim = a.n_avail - tile_lo, wl = (uint64_t)tid * 64u;
                const uint64_t dh = wl >= lim ? 0ull : (wl + 64u <= lim ? c.h : (c.h & ((1ull << (lim - wl)) - 1ull)));
                const uint64_t dw = __ballot(dh != 0ull);
                if (dw) {
                    const uint32_t lw2 = 63u - (uint32_t)__clzll((long long)dw);
                    const uint64_t hl = rl64(dh, lw2);
                    const uint64_t sstar = (uint64_t)lw2 * 64u + 63u - (uint32_t)__clzll((long long)hl);
                    if (tid == 0 && sstar < own_lim) atomicOr(&a.stats->flags, LIME_FLAG_HALO);
                }
            }
        }
        __syncthreads();
        if (MODE == 0 && a.ablate != 4 && a.ablate != 3) { phase_b<EBWT>(L, a); __syncthreads(); }
        if (tid == 0) {
            if (MODE == 1) a.tile_cnt[tile] = L.cnt;
            if (MODE != 2 && L.cnt) atomicAdd(&a.stats->n_clusters, (unsigned long long)L.cnt);
            if (MODE != 2 && L.maxlen) atomicMax(&a.stats->max_len, L.maxlen);
            if (MODE == 0 && L.upd) atomicAdd(&a.stats->n_updates, (unsigned long long)L.upd);
        }
        if (MODE == 2) {
            // ordered emission: rank of each accepted head inside the tile
            if (tid < 64u) {
                uint32_t cN = (uint32_t)__popcll(L.A[tid]);
                uint32_t x = cN;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)tid >= d) x += y; }
                L.apre[tid] = x - cN;
            }
            __syncthreads();
            const uint64_t base = a.tile_off[tile];
            const uint32_t wave = tid >> 6, lane = tid & 63u;
            for (uint32_t j = 0; j < WPW; ++j) {
                const uint32_t w = wave * WPW + j;
                const uint64_t am = L.A[w];
                if (!((am >> lane) & 1ull)) continue;
                const uint32_t s = w * 64u + lane;
                const uint64_t ha = (lane == 63u) ? 0ull : (L.H[w] & (~0ull << (lane + 1u)));
                const uint32_t e = ha ? w * 64u + (uint32_t)__builtin_ctzll(ha) : L.esuf[w];
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
        }
        __syncthreads();
        if (next >= a.n_tiles) break;
        if (a.ablate == 8) tile_load<EBWT>(regs, a, (uint64_t)next * TILE);
        tile = next;
    }
}

// =========================================================================================
// k_resolve: closes the segment that leaves each tile from the summaries of the tiles after
// it (the reference's straddle loop, ClusterLCP.cpp:246-264, and EOF closure :244-245).
// One thread per tile.  MODE 0: push to the small / big score lists; MODE 1: record for emit.
// =========================================================================================
template <int MODE>
__global__ __launch_bounds__(256) void k_resolve(ScanArgs a)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
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
// LIST_BATCH clusters (each <= SMALL_MAX long) side by side into LDS, makes one work item per
// read position and runs the same phase B as the tile kernel; longer clusters are pushed to
// the big list.  `count_ptr` (device) or `count` gives the number of records.
// =========================================================================================
template <int EBWT>
__global__ __launch_bounds__(WGSZ) void k_score_list(ScanArgs a, const lime_cluster_t *list,
                                                     const uint32_t *count_ptr, uint64_t count,
                                                     uint32_t cap)
{
    __shared__ TileLds L;
    __shared__ uint32_t c_off[LIST_BATCH + 1];
    __shared__ uint64_t c_ps[LIST_BATCH];
    __shared__ uint32_t w_tot[LIST_BATCH / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t n_list = count_ptr ? (uint64_t)(*count_ptr < cap ? *count_ptr : cap) : count;
    const uint64_t n_batches = (n_list + LIST_BATCH - 1) / LIST_BATCH;
    for (uint64_t b = blockIdx.x; b < n_batches; b += gridDim.x) {
        lds_reset(L);
        uint32_t len32 = 0, incl = 0;
        if (tid < LIST_BATCH) {
            const uint64_t c = b * LIST_BATCH + tid;
            uint64_t ps = 0, len = 0;
            if (c < n_list) { ps = list[c].pStart; len = list[c].len; }
            const bool bad = (len > LIME_MAX_CLUSTER) || (ps > a.n_avail) || (len > a.n_avail - ps);
            if (bad) { atomicOr(&a.stats->flags, len > LIME_MAX_CLUSTER ? LIME_FLAG_MAXLEN : LIME_FLAG_BADCLUSTER); len = 0; }
            if (len > SMALL_MAX) {
                uint32_t k = atomicAdd(&a.stats->n_big, 1u);
                if (k < a.big_cap) { a.big[k].pStart = ps; a.big[k].len = len; }
                len = 0;
            }
            if (len < 2u) len = 0;                 // a 0/1-symbol cluster cannot hold a read and a genome
            len32 = (uint32_t)len;
            incl = len32;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(incl, d); if ((int)lane >= d) incl += y; }
            if (lane == 63u) w_tot[tid >> 6] = incl;
            c_ps[tid] = ps;
        }
        __syncthreads();
        if (tid < LIST_BATCH) {
            uint32_t pre = 0;
            for (uint32_t k = 0; k < (tid >> 6); ++k) pre += w_tot[k];
            c_off[tid] = pre + incl - len32;
            if (tid == LIST_BATCH - 1) c_off[LIST_BATCH] = pre + incl;
        }
        __syncthreads();
        const uint32_t total = c_off[LIST_BATCH];
        for (uint32_t i = tid; i < total; i += WGSZ) {
            uint32_t lo = 0, hi = LIST_BATCH - 1;       // last cluster with c_off <= i
            while (lo < hi) { uint32_t mid = (lo + hi + 1u) >> 1; if (c_off[mid] <= i) lo = mid; else hi = mid - 1u; }
            const uint64_t g = c_ps[lo] + (i - c_off[lo]);
            const uint32_t d = a.da[g];
            uint32_t f = EBWT ? sym_index(a.ebwt[g]) : 0u;
            f |= (d < a.n_reads) ? F_READ : F_GEN;
            L.da[i] = d; L.fl[i] = (uint8_t)f;
        }
        if (tid < LIST_BATCH && len32) {
            const uint16_t item = (uint16_t)(c_off[tid] | ((len32 - 1u) << 12));
            if (len32 <= 4u) L.listA[atomicAdd(&L.nA, 1u)] = item;
            else if (len32 <= 8u) L.listB[atomicAdd(&L.nB, 1u)] = item;
            else L.listC[atomicAdd(&L.nC, 1u)] = item;
        }
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
    const dim3 grid((a.n_tiles + 255) / 256), block(256);
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
    uint64_t blocks = ((uint64_t)n_reads + (WGSZ / 64) - 1) / (WGSZ / 64);
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
