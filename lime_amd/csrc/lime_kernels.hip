// lime_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for LiME's hot path:
// alpha-cluster detection over lcp/da (reference: src/ClusterLCP.cpp:140-283) and per-cluster
// read x genome similarity accumulation over ebwt/da (reference: src/ClusterBWT_DA.cpp:256-358).
//
// Design: the unit of work is a POSITION; clusters are segments delimited by
// head(i) := lcp[i] < alpha.  Every WAVE is an independent worker over 1024-position windows (+16
// positions of read-ahead): lane-strided dword loads (register j of lane l = position 64 j + l),
// the next window's loads in flight while the current one is worked on, no workgroup barrier in
// the loop; the waves of a workgroup (one per CU) draw their windows from one LDS counter.  Per
// window: head / read bits as wave ballots (= mask words), acceptance of the segments by carry-ripple
// arithmetic on 16-bit chunks, a list of the accepted clusters' positions in LDS, clusters of 2..4
// symbols one per lane, 5..16 by fixed lane groups, 17..64 one at a time by the whole wave; table
// updates are queued in LDS and leave either as compare-and-swaps on the packed byte cells (exact
// modulo 256 like the reference's unsigned char) or as 4-byte records that later kernels bin and
// apply through LDS.  Segments still open after the read-ahead are closed by k_resolve_open /
// k_resolve; clusters longer than 64 go to a one-workgroup-per-cluster hash kernel.  Integer / byte
// work only: no MFMA, HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include "lime_device.h"
#include "lime_kernels.h"

namespace lime {

// ---- flag byte kept per staged position -------------------------------------------------
constexpr uint32_t F_SYM = 0x0F;
constexpr uint32_t F_READ = 0x20;
constexpr uint32_t F_GEN = 0x40;

constexpr uint32_t MID_MAX = 64;      // clusters of SMALL_MAX+1 .. MID_MAX symbols: scored by the scan as one 64-lane group each
constexpr uint32_t QCAP = 320;        // pending table updates per wave (a batch of 64 small clusters adds <= 256)
// queue entry: genome | t << T_SHIFT (lime_kernels.h): 7 bits of t; a cluster scored in the scan has <= MID_MAX symbols, so t <= MID_MAX / 2
static_assert(MID_MAX / 2u < (1u << (32u - T_SHIFT)), "the queue's t field must hold the largest in-scan pair score");
constexpr uint32_t CAP_A = 256;       // clusters (2..SMALL_MAX symbols) a window can own
constexpr uint32_t CAP_D = 256;       // of those, clusters with a repeated document (general routine)

// LDS is written and read through differently typed pointers (bytes as u16/u64, words as uint4):
// these may_alias types keep the compiler from reordering such accesses under type-based aliasing.
typedef volatile uint8_t __attribute__((address_space(3))) lds_vu8;
typedef volatile uint32_t __attribute__((address_space(3))) lds_vu32;
typedef uint16_t __attribute__((may_alias)) u16a;
typedef uint64_t __attribute__((may_alias)) u64a;
typedef uint4 __attribute__((may_alias)) u4a;

// LDS of ONE wave: the staged window and its work lists.  `da`/`fl` are sized by the kernel.
template <uint32_t NPOS>
struct alignas(16) WaveLds {
    uint32_t da[NPOS];
    uint8_t fl[NPOS];
    alignas(8) uint8_t hb[NPOS / 8 + 8];   // head bit of every staged position of the scan, same layout as rb (read as 64-bit words)
    alignas(8) uint8_t rb[NPOS / 8 + 8];   // read bit of every staged position (byte k = positions 8k..8k+7)
    uint16_t listM[WIN / 5 + 4];      // scan: this window's clusters of 5..SMALL_MAX symbols (start | (len-1) << 12)
    uint32_t m_tstart[64];            //       first pair-task of each of them
    uint8_t m_flag[64], m_dup[64];
    alignas(8) uint64_t asw[NW];                 // scan: per mask word, heads of the clusters scored in the window
    uint32_t prew[NW];                 //       and how many such heads the words before hold
    uint16_t listA[CAP_A], listD[CAP_D];   // entry: start | (len-1) << 12
    uint32_t q_read[QCAP], q_gen[QCAP];
};

// timing experiments (tools/quick.sh): a build with -DLIME_ABLATE_BUILD cuts the scan after phase k when the
// environment says LIME_ABLATE=k (results invalid); the release library has no such switch
#ifdef LIME_ABLATE_BUILD
#define ABL(k) (a.ablate == (k))
#else
#define ABL(k) false
#endif

#ifdef LIME_WALL_TIMING      // debug build: start and end wall clock (100 MHz) of every wave of the last scan
__device__ uint64_t g_wall[2 * 8192];
extern "C" int lime_debug_wall(uint64_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wall), sizeof(g_wall)); }
__device__ uint32_t g_winmark[1u << 20];       // which wave (+ 1) took window w of the last scan
extern "C" int lime_debug_winmark(uint32_t *out)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_winmark), sizeof(g_winmark));
    void *p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_winmark)); (void)hipMemset(p, 0, sizeof(g_winmark));
    return rc;
}
#endif
#ifdef LIME_PHASE_TIMING     // debug build: per-wave cycle counts of the scan's phases, printed by a few waves
#define PT_DECL uint64_t pt_t = __builtin_readcyclecounter(), pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_m[4] = {0, 0, 0, 0}; uint32_t pt_nwin = 0;
#define PT(i) { const uint64_t n_ = __builtin_readcyclecounter(); pt_acc[i] += n_ - pt_t; pt_t = n_; }
#define PT_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif defined(LIME_MARK)     // ISA reading aid: phase borders as comments in the assembly (tools/isa_count.py)
#define PT_DECL
#define PT(i) asm volatile("; LIMEMARK " #i);
#define PT_WAITVM
#else
#define PT_DECL
#define PT(i)
#define PT_WAITVM
#endif

// threads per workgroup of k_scan and waves per SIMD it is compiled for: ScanCfg in lime_kernels.h (one workgroup per CU:
// EBWT = 0: 16 waves = 4 per SIMD; EBWT = 1: 12 waves = 3 per SIMD); the LDS of one wave is kept small: it bounds them
constexpr uint32_t DUP_SLOTS = 8;     // clusters with a repeated document a wave of k_scan holds before scoring them
constexpr uint32_t DUP_SLOTS_E = 4;   // ... with symbols (EBWT=1: 16 waves per CU, the LDS is short): one flush of four 16-lane groups
constexpr uint32_t QCAP_SCAN = 286;   // >= the 256 hits one batch of 64 clusters of <= 4 symbols can add + the 2 x 15 entries a binned drain leaves behind
constexpr uint32_t QCAP_SCAN_SHORT = 160;   // EBWT=1 with records (16 waves per CU): a batch that would not fit is emitted in two halves (score_small3)

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// The kernel's ScanArgs (always its first argument) re-read from the kernarg segment at the point of use: fields that only
// rare paths need (counters, flags, the long-cluster list ...) then cost a scalar load there instead of SGPRs held through
// the whole window loop -- the scan kernels were 2..40 SGPRs over budget and spilled them into VGPR lanes.
__device__ __forceinline__ const ScanArgs &cold(const ScanArgs &)
{
    const ScanArgs __attribute__((address_space(4))) *p = (const ScanArgs __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                               // opaque: not merged with the by-value copy, not hoisted
    return *(const ScanArgs *)p;
}
__device__ __forceinline__ uint64_t brev64(uint64_t x) { return __builtin_bitreverse64(x); }
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// v with lane L replaced by the wave-uniform value s (this compiler has no v_writelane builtin).  On gfx940/gfx950 a
// vector instruction that reads an SGPR / VCC written by the vector instruction just before it -- here the v_cmp whose
// ballot is s -- needs two wait states.  The compiler's hazard recognizer inserts them for instructions it knows; inline
// assembly is opaque to it, and whenever the scheduler happened to put a bare `v_writelane` right behind its v_cmp it read
// the PREVIOUS ballot: round 2's "wrong cluster counts" of the build with the runtime update-path flag, and round 3's of the
// first lean EBWT=1 binned scan (`v_cmp_gt_u32 vcc, ..` / `v_writelane_b32 v2, vcc_lo, 1` back to back).
// So the wait states are part of the assembly: `s_nop 1` (two wait states) in front, and the four writes of a mask word
// share one statement and one nop.
template <int L> __device__ __forceinline__ uint32_t write_lane(uint32_t v, uint32_t s)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(L));
    return v;
}
template <int L> __device__ __forceinline__ void write_lane4(uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &v3,
                                                            uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %4, %8\n\tv_writelane_b32 %1, %5, %8\n\tv_writelane_b32 %2, %6, %8\n\tv_writelane_b32 %3, %7, %8"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "n"(L));
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l)
{
    return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l);
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int l)
{
    return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), l) << 32) | (uint32_t)__shfl((int)(uint32_t)v, l);
}
// number of set bits of the wave mask m in lanes below this one (v_mbcnt_lo/hi: two instructions)
__device__ __forceinline__ uint32_t rank_in(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// 32 mask bits starting at bit `pos` of a bit array in LDS (8-byte aligned, at least (pos >> 5) + 2 words long):
// two aligned words and one v_alignbit
typedef uint32_t __attribute__((may_alias)) u32a;
__device__ __forceinline__ uint32_t bits_at(const uint8_t *bits, uint32_t pos)
{
    const u32a *w = reinterpret_cast<const u32a *>(bits) + (pos >> 5);
    return __builtin_amdgcn_alignbit(w[1], w[0], pos & 31u);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// byte -> symbol index, and symbol index -> set of symbol indices it scores 1 against
// (iupac_match); filled once per workgroup
// pairlut[rmask | (len-1) << 4] for a cluster of len = 2..4 symbols with read bits rmask: bit k = position
// pair k of (0,1)(0,2)(0,3)(1,2)(1,3)(2,3) joins a read with a genome inside the cluster, bit 8+k =
// the pair's FIRST position is the read
// pairs[rmask | (len-1) << 4]: the read x genome pairs of a cluster of len = 2..4 symbols with read bits rmask, as a
// list of up to four slots of 7 bits (read position | genome position << 2 | number of the position pair << 4) and
// their count in bits 28..30
struct WgTables { uint8_t symidx[256]; uint16_t compat[16]; uint16_t pairlut[64]; uint16_t compatb[256]; uint32_t pairs[64]; };   // compatb[byte] = compat[symidx[byte]]

__device__ __forceinline__ void tables_init(WgTables &T)
{
    const uint32_t t = threadIdx.x;
    for (uint32_t b = t; b < 256u; b += blockDim.x) {
        const uint32_t si = sym_index(b);
        uint32_t m = 0;
        for (uint32_t k = 0; k < 16u; ++k) m |= iupac_match(si, k) << k;
        T.symidx[b] = (uint8_t)si; T.compatb[b] = (uint16_t)m;
    }
    if (t < 16u) {
        uint32_t m = 0;
        for (uint32_t b = 0; b < 16u; ++b) m |= iupac_match(t, b) << b;
        T.compat[t] = (uint16_t)m;
    }
    for (uint32_t e = t; e < 64u; e += blockDim.x) {
        const uint32_t len = (e >> 4) + 1u, vm = (1u << len) - 1u, r = e & 15u & vm, g = ~r & vm;
        uint32_t v = 0, k = 0;
        for (uint32_t i = 0; i < 4u; ++i)
            for (uint32_t j = i + 1u; j < 4u; ++j, ++k) {
                const uint32_t ri = (r >> i) & 1u, rj = (r >> j) & 1u, gi = (g >> i) & 1u, gj = (g >> j) & 1u;
                v |= ((ri & gj) | (gi & rj)) << k;
                v |= ri << (8u + k);
            }
        T.pairlut[e] = (uint16_t)v;
        uint32_t pv = 0, np = 0;
        k = 0;
        for (uint32_t i = 0; i < 4u; ++i)
            for (uint32_t j = i + 1u; j < 4u; ++j, ++k) {
                const uint32_t ri = (r >> i) & 1u, rj = (r >> j) & 1u, gi = (g >> i) & 1u, gj = (g >> j) & 1u;
                if ((ri & gj) | (gi & rj)) { pv |= ((ri ? i : j) | ((ri ? j : i) << 2) | (k << 4)) << (7u * np); ++np; }
            }
        T.pairs[e] = pv | (np << 28);
    }
    __syncthreads();
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

// ---- table updates of a wave: queued in its LDS ring, applied together so that the round
// trips of the compare-and-swaps overlap instead of following one another -----------------
struct UpdQueue { uint32_t *qr, *qg; uint32_t n, cap;
    // split-phase mode (k_scan): the compare-and-swaps of a drain are only ISSUED; their results
    // are looked at by the next drain, so a wave never waits for the round trip to the table.
    // Lane l owns in-flight slots 64 j + l (j < 4): entry kept in fr/fg, value the CAS expected
    // and value it returned in registers.
    bool async = false;
    uint32_t *fr = nullptr, *fg = nullptr, *fe = nullptr;   // entry (read, genome | t) and the word value its CAS expected
    uint32_t f_old[4], f_pend = 0;               // f_pend: bit j = slot j occupied, bit 4+j = its word was only LOADED so far
    uint32_t nj = 4;                             // in-flight slots per lane (a constant of the kernel: 4, or 2 where LDS is short)
    // dense tables: once 1 in 4 of the first tries (which expect an empty word) has lost, new entries first load
    // their word (a load is much cheaper than a lost compare-and-swap) and try with what they saw one drain later
    uint32_t f_first = 0, f_lost = 0; bool load_first = false;
    // binned mode (ScanArgs::upd_mode): drains append (cell, t) records to the wave's region of the pool and count
    // them per table bin in the workgroup's LDS histogram; no table access from the scan at all
    bool binned = false; uint32_t *out = nullptr; lds_vu32 *sub_n = nullptr; uint32_t *hist = nullptr;   // out: the wave's n_sub sub-regions; sub_n: records in each (LDS)
    uint32_t *lbuf = nullptr; lds_vu32 *lfill = nullptr;   // one or two sub-regions: finished records wait here (LBUF per sub-region, lfill[s] of them) until they fill 64-byte lines
    // direct mode (round 6; k_scan<.,0,2>: tables of one or two sub-regions): the scorers write FINISHED 4-byte records -- qr[] holds sub-region 0's, qg[]
    // sub-region 1's, n / n1 of them -- and a flush stores their whole 64-byte lines; no queue entry is looked at twice
    bool direct = false;
    uint32_t n1 = 0, base0 = 0, base1 = 0;                 // records waiting in qg[]; records stored so far per sub-region (wave-uniform)
    uint32_t sub_rb = 0xFFFFFFFFu, sub_gb = 0;             // where sub-region 1 begins (read, genome); sub_rb = ~0: one sub-region
    __device__ __forceinline__ bool two() const { return sub_rb != 0xFFFFFFFFu; }
    uint32_t bad = 0;                                      // per lane: a genome index beyond the table was met (LIME_FLAG_DOCID at the next flush)
#ifdef LIME_PHASE_TIMING
    uint64_t t_drain = 0; uint32_t n_drain = 0;
#endif
};

// The wave's region of the record pool (n_sub sub-regions of cap_w records), recomputed where it is needed from the kernel's arguments and the wave's number
__device__ __forceinline__ uint32_t *pool_of(const ScanArgs &ca, uint32_t n_sub, uint32_t cap_w)
{
    const uint32_t wave_gid = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    return ca.pool + (size_t)wave_gid * n_sub * cap_w;
}

// Split-phase drain: (1) settle the slots issued last time: a CAS that found the expected word is
// done, one that lost keeps its slot with the word it saw, a slot that only loaded its word now knows
// what to expect; (2) free slots take entries from the queue's tail; (3) every occupied slot issues
// its CAS (or, for a new entry in load-first mode, the load of its word).  Entries that found no free
// slot stay queued (q.n > 0 afterwards): callers loop while they need more room.
__device__ __forceinline__ void drain_async(UpdQueue &q, const ScanArgs &a)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if ((uint32_t)j >= q.nj) break;
        const bool pend = (q.f_pend >> j) & 1u, fresh = (q.f_pend >> (4 + j)) & 1u;
        const uint32_t e = pend ? q.fe[64u * (uint32_t)j + lane] : 0u;
        const bool tried = pend && !fresh, lost = tried && q.f_old[j] != e;
        if (j == 0 && !q.load_first) {                     // statistics of the "expect an empty word" first tries (slot 0 as a sample)
            const uint64_t mf = __ballot(tried && e == 0u);
            q.f_first += (uint32_t)__popcll(mf); q.f_lost += (uint32_t)__popcll(mf & __ballot(lost));
        }
        if (tried && !lost) q.f_pend &= ~(1u << j);
        if (fresh || lost) { q.fe[64u * (uint32_t)j + lane] = q.f_old[j]; q.f_pend &= ~(16u << j); }
    }
    if (!q.load_first && q.f_first >= 64u && 4u * q.f_lost >= q.f_first) q.load_first = true;
    uint32_t n = q.n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if ((uint32_t)j >= q.nj) break;
        const bool fre = !((q.f_pend >> j) & 1u);
        const uint64_t m = __ballot(fre);
        const uint32_t r = (uint32_t)__popcll(m & lt), c = (uint32_t)__popcll(m);
        if (fre && r < n) {
            const uint32_t k = n - 1u - r;
            const uint32_t gt = q.qg[k];
            if ((gt & ((1u << T_SHIFT) - 1u)) >= a.n_refs) atomicOr(&cold(a).stats->flags, LIME_FLAG_DOCID);   // dropped: see drain_bin
            else {
                q.fr[64u * (uint32_t)j + lane] = q.qr[k]; q.fg[64u * (uint32_t)j + lane] = gt;
                q.fe[64u * (uint32_t)j + lane] = 0u;
                q.f_pend |= (q.load_first ? 17u : 1u) << j;
            }
        }
        n -= c < n ? c : n;
    }
    q.n = n;
    if (ABL(5)) { q.f_pend = 0; return; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((uint32_t)j < q.nj && ((q.f_pend >> j) & 1u)) {
            const uint32_t gt = q.fg[64u * (uint32_t)j + lane];
            const uint64_t cell = (uint64_t)q.fr[64u * (uint32_t)j + lane] * a.n_refs + (gt & ((1u << T_SHIFT) - 1u));
            uint32_t *w = reinterpret_cast<uint32_t *>(a.sim + (cell & ~3ull));
            if ((q.f_pend >> (4 + j)) & 1u) {
                q.f_old[j] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // device-coherent: not a stale L2 line
            } else {
                const uint32_t sh = (uint32_t)(cell & 3ull) * 8u, e = q.fe[64u * (uint32_t)j + lane];
                const uint32_t b = ((e >> sh) + (gt >> T_SHIFT)) & 255u;
                q.f_old[j] = atomicCAS(w, e, (e & ~(255u << sh)) | (b << sh));
            }
        }
}

// Tables of one or two sub-regions (up to 8 GB): the queue's entries become finished 4-byte records in a small LDS
// buffer per sub-region, and only WHOLE 64-byte lines leave it (partial-line writes cost the N = 10^10 scan 5 %); up to 15
// records per sub-region wait there for the next drain, as records -- nothing is looked at twice (the first version
// kept them as queue entries: a counting loop, a compaction of the kept entries and their 64-bit cell arithmetic again in
// every drain, ~200 vector instructions per window).  `final`: the kernel's last drain writes the partial lines too.
constexpr uint32_t LBUF = 80;                                     // <= 15 waiting + 64 new records per sub-region
__device__ __forceinline__ void drain_lines(UpdQueue &q, const ScanArgs &a, bool final)
{
    const uint32_t lane = lane_id();
    const uint32_t n = q.n;
    const ScanArgs &ca = cold(a);                                 // the fields a drain needs are loaded here, not held through the window loop
    const uint32_t cap_w = ca.cap_w, bin_shift = ca.bin_shift, n_sub = ca.n_sub, sub_rb = ca.sub_rb, sub_gb = ca.sub_gb, n_refs = ca.n_refs;
    uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.lfill[0]), f1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.lfill[1]);
    const uint32_t hoff = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.sub_n[MAX_SUB]);
    auto flush = [&](uint32_t sub, uint32_t &f, uint32_t keep_mask) {      // keep_mask = 15: whole lines only; 0: everything
        const uint32_t nl = f & ~keep_mask;
        if (!nl) return;
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.sub_n[sub]);
        uint32_t *lb = q.lbuf + sub * LBUF;
        const uint32_t rec = lb[lane < nl ? lane : 0u];
        const uint32_t slot = base + lane;
        if (lane < nl && slot < cap_w && !ABL(7)) {               // a full sub-region only counts (sub_n): the pass is repeated with a larger pool
            const uint32_t bin = n_sub == 1u ? rec >> bin_shift : (rec >> bin_shift) | (sub << (32u - bin_shift));
            atomicAdd(&q.hist[hoff + bin], 1u);                   // the histogram counts exactly the records that are stored
            if (!ABL(6)) __builtin_nontemporal_store(rec, q.out + (size_t)sub * cap_w + slot);
        }
        const uint32_t rem = f - nl;
        const uint32_t mv = lb[lane < rem ? nl + lane : 0u];      // (all lanes read before any writes: one instruction each)
        if (lane < rem) lb[lane] = mv;
        if (lane == 0) q.sub_n[sub] = base + nl;
        f = rem;
    };
    for (uint32_t k0 = 0; k0 < n; k0 += 64u) {
        const uint32_t k = k0 + lane;
        const bool on = k < n;
        const uint32_t gt = q.qg[on ? k : 0u], rd = q.qr[on ? k : 0u];
        const uint32_t g = gt & (MAX_REFS - 1u);
        // the scan's fast emitters do not look at the document ids: a genome id beyond the table is caught here, on
        // full waves (the entry is dropped; the pass fails with LIME_ERR_DOCID)
        const bool bad = on && g >= n_refs;
        if (__ballot(bad)) { if (bad) atomicOr(&cold(a).stats->flags, LIME_FLAG_DOCID); }
        const uint32_t rec = rd * n_refs + g;                     // the cell's low 32 bits
        const bool hi = rd > sub_rb || (rd == sub_rb && g >= sub_gb);
        uint32_t left = (on && !bad) ? gt >> T_SHIFT : 0u;
        while (__ballot(left != 0u)) {                            // once, unless a pair scored more than 1
            const bool act = left != 0u;
            const uint64_t m1 = __ballot(act && hi), m0 = __ballot(act && !hi);
            if (act) q.lbuf[hi ? LBUF + f1 + rank_in(m1) : f0 + rank_in(m0)] = rec;
            f0 += (uint32_t)__popcll(m0); f1 += (uint32_t)__popcll(m1);
            left -= (uint32_t)act;
            flush(0u, f0, 15u);
            if (m1) flush(1u, f1, 15u);
        }
    }
    q.n = 0;
    if (final) { flush(0u, f0, 0u); flush(1u, f1, 0u); }
    if (lane == 0) { q.lfill[0] = f0; q.lfill[1] = f1; }
}

// Direct mode (round 6).  Ablation cuts showed what the queue -> record -> line-buffer copy of drain_lines costs beyond its stores: 223 of 1586 us on
// configs[2], 2.3 of 14.7 ms at N = 1e10, 61 of 251 us on the text workload (profiles/r06_scan_cuts.txt: the scan without ANY drain work runs at the
// memory floor, 1234 us) -- two LDS reads per entry, a quarter-rate multiply, two ballots and ranks, an LDS write, then the line buffer's read-back, and
// all of it in front of the next window's loads.  Here the scorers compute the record themselves (one multiply-add where they used to write two words)
// and write it where its 64-byte line is being gathered; what is left for the flush is: read 64 records, count them in the bin histogram, store.
__device__ __forceinline__ void flush_direct(UpdQueue &q, const ScanArgs &a, bool final)
{
    if (ABL(13)) return;
    if (ABL(14)) { q.n = 0; q.n1 = 0; return; }                      // (timing experiments: the scorers' share of the record work without the flush)
    const uint32_t lane = lane_id();
    const ScanArgs &ca = cold(a);
    const uint32_t cap_w = ca.cap_w, bin_shift = ca.bin_shift, n_sub = ca.n_sub, bin_lim = ca.n_bins;
    const uint32_t hoff = ((uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) / ca.prod_waves) * bin_lim;     // the producer group's part of the workgroup's histogram
    if (__ballot(q.bad != 0u)) { if (q.bad) atomicOr(&cold(a).stats->flags, LIME_FLAG_DOCID); q.bad = 0u; }
    uint32_t *const out = pool_of(ca, n_sub, cap_w);
#pragma unroll
    for (uint32_t sub = 0; sub < 2u; ++sub) {
        if (sub && !q.two()) break;
        const uint32_t n = sub ? q.n1 : q.n, nl = final ? n : n & ~15u;
        if (!nl) continue;                                           // wave-uniform
        uint32_t *buf = sub ? q.qg : q.qr;
        const uint32_t base = sub ? q.base1 : q.base0;
        for (uint32_t k0 = 0; k0 < nl; k0 += 64u) {
            const uint32_t k = k0 + lane;
            const bool on = k < nl;
            const uint32_t rec = buf[on ? k : 0u], slot = base + k;
            uint32_t bin = (rec >> bin_shift) | (sub << (32u - bin_shift));
            bin = bin < bin_lim ? bin : bin_lim - 1u;                // (only a pass that fails with LIME_ERR_DOCID can get here with a cell beyond the table)
            if (on && slot < cap_w && !ABL(7)) {                     // a full sub-region only counts: the pass is repeated with a larger pool
                atomicAdd(&q.hist[hoff + bin], 1u);                  // the histogram counts exactly the records that are stored
                if (!ABL(6)) __builtin_nontemporal_store(rec, out + (size_t)sub * cap_w + slot);
            }
        }
        const uint32_t rem = n - nl;                                 // < 16 <= nl: the two ranges below do not overlap
        const uint32_t mv = buf[lane < rem ? nl + lane : 0u];
        if (lane < rem) buf[lane] = mv;
        if (sub) { q.base1 = base + nl; q.n1 = rem; } else { q.base0 = base + nl; q.n = rem; }
    }
}

// direct mode: the lanes with `hit` append the record of (read rd, genome index gd), t = 1.  All 64 lanes call.
__device__ __forceinline__ void put_rec(UpdQueue &q, const ScanArgs &a, bool hit, uint32_t rd, uint32_t gd)
{
    if (ABL(13)) return;
    const uint32_t rec = rd * a.n_refs + gd;                         // the cell's low 32 bits
    q.bad |= (uint32_t)(hit && gd >= a.n_refs);
    if (!q.two()) {
        const uint64_t m = __ballot(hit);
        const uint32_t tot = (uint32_t)__popcll(m);
        while (q.n + tot > q.cap) flush_direct(q, a, false);
        if (hit) q.qr[q.n + rank_in(m)] = rec;
        q.n += tot;
    } else {
        const bool hi = rd > q.sub_rb || (rd == q.sub_rb && gd >= q.sub_gb);      // cell >= 2^32
        const uint64_t m1 = __ballot(hit && hi), m0 = __ballot(hit && !hi);
        const uint32_t t0 = (uint32_t)__popcll(m0), t1 = (uint32_t)__popcll(m1);
        while (q.n + t0 > q.cap || q.n1 + t1 > q.cap) flush_direct(q, a, false);
        if (hit) { if (hi) q.qg[q.n1 + rank_in(m1)] = rec; else q.qr[q.n + rank_in(m0)] = rec; }
        q.n += t0; q.n1 += t1;
    }
}

// Binned mode: the queue's entries become 4-byte pool records in the wave's own region: the LOW 32 bits of the cell; the
// high part picks one of the wave's n_sub sub-regions and the score is implicit (a pair that scores t > 1 -- repeated
// documents only -- leaves t records).  Every byte the scan stores costs its read stream dearly -- a gigabyte of appended
// records slows 8 GB of loads from 1.15 to 1.76 ms whatever the layout of the stores (tools/load_bench.hip) -- so the
// records are as short as they can be.  Records beyond a sub-region's capacity are only counted (the host repeats the pass
// with a larger pool: LIME_FLAG_POOL_FULL).  One or two sub-regions (tables up to 8 GB): drain_lines above; more: here,
// 64 records at a time as they come.
// `whole`: every entry leaves (the window's top, the kernel's end); else -- a scorer needs room in the middle of a window -- only the LAST 64: a full
// batch (the order of the records is free), instead of a full one and the few entries behind it every time.
__device__ __forceinline__ void drain_bin(UpdQueue &q, const ScanArgs &a, bool final = false, bool whole = true)
{
    if (cold(a).n_sub <= 2u) { drain_lines(q, a, final); return; }         // (re-read: a flag held through the window loop is a pair of SGPRs)
    // Three and more sub-regions.  Until round 6 the records of such a table left "as they came": every 64 queue entries split three to eight ways by a
    // loop over the sub-regions (a ballot, a rank, an LDS read and write of the sub-region's count each), so nearly every store was a part of a
    // 64-byte line -- the stores alone were 2.65 of the 16.7 ms of the scan at configs[4]'s shape, and where records are dense the loop was a third of
    // the scan (5.3 of 20.7 ms on the clustered generator; profiles/r06_scan_cuts.txt).  Now there is no loop: a lane takes its record's place in its
    // sub-region with ONE LDS atomic on the sub-region's running count (sub_n[]: the hardware serialises the lanes of one address), and only whole
    // lines leave: a record whose place lies in a line that this batch completes is stored from its register, the others wait in LDS (lbuf[16 s + i]:
    // record i of sub-region s's open line), and the records that waited in a line completed now are stored by the lanes 16 s + i that read them
    // back at the top (two store instructions into one line, back to back: they meet in L2).  The order of a sub-region's records is whatever the
    // LDS made it -- the table is a sum.
    const uint32_t lane = lane_id();
    const ScanArgs &ca = cold(a);
    const uint32_t n = q.n, n_sub = ca.n_sub, cap_w = ca.cap_w, n_refs = ca.n_refs, bin_shift = ca.bin_shift;
    const uint32_t hoff = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.sub_n[MAX_SUB]);
    uint32_t *const out = pool_of(ca, n_sub, cap_w);
    const uint32_t ws = lane >> 4, wi = lane & 15u;               // this lane's part in storing waiting records: record wi of sub-region ws (and ws + 4)
    // the waiting records of sub-regions ws / ws + 4 whose line the batch completed (counts c0 before, c1 after) leave
    auto store_waiting = [&](uint32_t sub, uint32_t c0, uint32_t c1, uint32_t wrec) {
        const uint32_t l0 = c0 & ~15u;
        if (sub < n_sub && (c1 & ~15u) != l0 && wi < (c0 & 15u) && l0 + wi < cap_w && !ABL(7)) {     // a full sub-region only counts: the pass is repeated with a larger pool
            atomicAdd(&q.hist[hoff + ((wrec >> bin_shift) | (sub << (32u - bin_shift)))], 1u);      // the histogram counts exactly the records that are stored
            if (!ABL(6)) out[(size_t)sub * cap_w + l0 + wi] = wrec;       // (a plain store, like the other piece of its line below: a non-temporal store of less than a line is a read-modify-write at the memory, 4.5 x a whole line's -- plain, the two pieces meet in L2)
        }
    };
    const uint32_t first = whole || n <= 64u ? 0u : n - 64u;      // entries [first, n) leave
    for (uint32_t k0 = first; k0 < n; k0 += 64u) {
        const uint32_t k = k0 + lane;
        const bool on = k < n;
        const uint32_t gt = q.qg[on ? k : 0u], rd = q.qr[on ? k : 0u];
        // the scan's fast emitters do not look at the document ids: a genome id beyond the table is caught here, on
        // full waves (the entry is dropped; the pass fails with LIME_ERR_DOCID)
        const uint64_t cell = (uint64_t)rd * n_refs + (gt & (MAX_REFS - 1u));
        const uint32_t hi = (uint32_t)(cell >> 32), rec = (uint32_t)cell;
        const bool bad = on && ((gt & (MAX_REFS - 1u)) >= n_refs || hi >= n_sub);
        if (__ballot(bad)) { if (bad) atomicOr(&cold(a).stats->flags, LIME_FLAG_DOCID); }
        uint32_t left = (on && !bad) ? gt >> T_SHIFT : 0u;
        while (__ballot(left != 0u)) {                            // once, unless a pair scored more than 1
            const bool act = left != 0u;
            const uint32_t wv0 = q.lbuf[lane], c00 = q.sub_n[ws];
            uint32_t wv1 = 0, c04 = 0;
            if (n_sub > 4u) { wv1 = q.lbuf[64u + lane]; c04 = q.sub_n[4u + ws]; }                   // wave-uniform
            uint32_t slot = 0;
            if (act) slot = __hip_atomic_fetch_add(&q.sub_n[hi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t c1h = q.sub_n[act ? hi : 0u], c10 = q.sub_n[ws];
            const uint32_t lim = c1h & ~15u;                      // the whole lines of this lane's sub-region end here
            if (act) {
                if (slot >= lim) q.lbuf[16u * hi + (slot & 15u)] = rec;
                else if (slot < cap_w && !ABL(7)) {
                    atomicAdd(&q.hist[hoff + (uint32_t)(cell >> bin_shift)], 1u);
                    if (!ABL(6)) out[(size_t)hi * cap_w + slot] = rec;
                }
            }
            store_waiting(ws, c00, c10, wv0);
            if (n_sub > 4u) store_waiting(4u + ws, c04, q.sub_n[4u + ws], wv1);
            left -= (uint32_t)act;
        }
    }
    q.n = first;
    if (final) {                                                  // the kernel's last drain: the open lines too (c1 = c0 + 16: "completed")
        const uint32_t c00 = q.sub_n[ws], wv0 = q.lbuf[lane];
        store_waiting(ws, c00, c00 + 16u, wv0);
        if (n_sub > 4u) { const uint32_t c04 = q.sub_n[4u + ws], wv1 = q.lbuf[64u + lane]; store_waiting(4u + ws, c04, c04 + 16u, wv1); }
    }
}
__device__ __forceinline__ void drain(UpdQueue &q, const ScanArgs &a)
{
    if (ABL(13)) { q.n = 0; q.n1 = 0; return; }            // (timing experiments: what ALL of the drains' work costs -- the entries are thrown away)
    if (q.direct) { flush_direct(q, a, false); return; }
    if (q.async) { if (q.binned) drain_bin(q, a, false, false); else drain_async(q, a); return; }     // both flags are compile-time constants of the kernel
#ifdef LIME_PHASE_TIMING
    const uint64_t t0 = __builtin_readcyclecounter();
#endif
    // four queue entries per lane and round: their compare-and-swaps are issued together (first
    // try: word still zero, tables are sparse) and only the ones that lost are tried again
    if (!ABL(5))
        for (uint32_t k0 = 0; k0 < q.n; k0 += 256u) {
            uint32_t *w[4], sh[4], t[4], expect[4], pend = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t k = k0 + 64u * (uint32_t)j + lane_id();
                const bool on = k < q.n;
                const uint32_t gt = q.qg[on ? k : 0u];
                const uint64_t cell = (uint64_t)q.qr[on ? k : 0u] * a.n_refs + (gt & ((1u << T_SHIFT) - 1u));
                w[j] = reinterpret_cast<uint32_t *>(a.sim + (cell & ~3ull));
                sh[j] = (uint32_t)(cell & 3ull) * 8u; t[j] = gt >> T_SHIFT; expect[j] = 0u;
                pend |= (uint32_t)on << j;
            }
            while (__ballot(pend != 0u)) {
                uint32_t old[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((pend >> j) & 1u) {
                        const uint32_t b = ((expect[j] >> sh[j]) + t[j]) & 255u;
                        old[j] = atomicCAS(w[j], expect[j], (expect[j] & ~(255u << sh[j])) | (b << sh[j]));
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((pend >> j) & 1u) {
                        if (old[j] == expect[j]) pend &= ~(1u << j);
                        else expect[j] = old[j];
                    }
            }
        }
    q.n = 0;
#ifdef LIME_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    q.t_drain += __builtin_readcyclecounter() - t0; ++q.n_drain;
#endif
}

// every active lane may add one update; all 64 lanes must call (wave ballots inside)
__device__ __forceinline__ uint32_t emit(UpdQueue &q, const ScanArgs &a, bool on, uint32_t rdoc, uint32_t gdoc, uint32_t t)
{
    if (!q.direct) while (q.n + 64u > q.cap) drain(q, a);
    const uint32_t g = gdoc - a.n_reads;
    const bool bad = on && (g >= a.n_refs || rdoc >= a.n_reads);
    if (__ballot(bad)) { if (bad) atomicOr(&cold(a).stats->flags, LIME_FLAG_DOCID); }
    on = on && !bad;
    if (q.direct) {                                        // a score of t = t records (t > 1: repeated documents only)
        uint32_t left = on ? t : 0u;
        while (__ballot(left != 0u)) { put_rec(q, a, left != 0u, rdoc, g); left -= (uint32_t)(left != 0u); }
        return on ? 1u : 0u;
    }
    const uint64_t m = __ballot(on);
    if (on) {
        const uint32_t slot = q.n + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
        q.qr[slot] = rdoc; q.qg[slot] = g | (t << T_SHIFT);
    }
    q.n += (uint32_t)__popcll(m);
    return on ? 1u : 0u;
}

// ---- general score of short clusters by lane groups ------------------------------------------
// G lanes per cluster, element i on sub-lane i; XOR-ing the sub-lane with 1..R (R + 1 = the power
// of two covering the longest cluster of the wave) shows every lane every other element of its
// cluster.  Pass 1: is this element the first of its document, and the document's count / 16-bin
// histogram over the cluster (counts <= 64: no wrap, no saturation).  Pass 2: every
// first-occurrence read meets every first-occurrence genome once (pair_score;
// ClusterBWT_DA.cpp:107-190 / :192-252).
// the clusters of one wave iteration: element `sub` of the lane's group holds document d and ebwt
// byte bb (sub < len; len = 0: no cluster in this group)
template <int EBWT, int G>
__device__ __forceinline__ uint32_t group_score(const ScanArgs &a, const WgTables &T, UpdQueue &qu,
                                                uint32_t d, uint32_t bb, uint32_t len)
{
    const uint32_t sub = lane_id() & (G - 1u);
    uint32_t acc_upd = 0;
    // rounds: sub-lane ^ r stays inside the first R+1 sub-lanes, which hold the whole cluster
    uint32_t R = 1u;
#pragma unroll
    for (uint32_t p2 = 2u; p2 < (uint32_t)G; p2 <<= 1) if (__ballot(len > p2)) R = 2u * p2 - 1u;
    const bool have = sub < len;
    const uint32_t sy = (EBWT && have) ? T.symidx[bb] : 0u;
    const bool isr = have && d < a.n_reads;
    const uint32_t f = have ? (sy | (isr ? F_READ : F_GEN)) : 0u;
    uint32_t earlier = 0, cnt = have ? 1u : 0u, hs[4] = {0u, 0u, 0u, 0u};
    if (EBWT) hist_add(hs, sy, (uint32_t)have);
#pragma unroll 1
    for (uint32_t r = 1; r <= R; ++r) {
        const uint32_t pd = __shfl_xor(d, (int)r), pf = __shfl_xor(f, (int)r);
        const uint32_t same = (uint32_t)(f && pf && pd == d);
        earlier |= same & (uint32_t)((sub ^ r) < sub);
        cnt += same;
        if (EBWT) hist_add(hs, pf & F_SYM, same);
    }
    const uint32_t lead = (uint32_t)(have && !earlier);
#pragma unroll 1
    for (uint32_t r = 1; r <= R; ++r) {
        const uint32_t pd = __shfl_xor(d, (int)r), pf = __shfl_xor(f, (int)r), pl = __shfl_xor(lead, (int)r), pc = __shfl_xor(cnt, (int)r);
        uint32_t ph[4] = {0u, 0u, 0u, 0u};
        if (EBWT) { ph[0] = __shfl_xor(hs[0], (int)r); ph[1] = __shfl_xor(hs[1], (int)r); ph[2] = __shfl_xor(hs[2], (int)r); ph[3] = __shfl_xor(hs[3], (int)r); }
        const bool pair = lead && isr && pl && (pf & F_GEN);
        if (__ballot(pair) == 0ull) continue;
        uint32_t t = cnt < pc ? cnt : pc;
        if (EBWT) {
            uint32_t h2[4], p2[4];           // only real pairs go through the (possibly slow) score
#pragma unroll
            for (int k = 0; k < 4; ++k) { h2[k] = pair ? hs[k] : 0u; p2[k] = pair ? ph[k] : 0u; }
            t = pair_score(h2, p2);
        }
        acc_upd += emit(qu, a, pair && t, d, pd, t);
    }
    return acc_upd;
}


// ---- clusters with a repeated document, kept by the wave that found them: copies of up to
// DUP_SLOTS clusters (<= SMALL_MAX elements each) in LDS, scored four at a time by 16-lane groups
// when enough have gathered (and at the end).  A full store is scored on the spot, so no cluster
// ever leaves the wave.
template <int EBWT, typename LDS>
__device__ __forceinline__ uint32_t dup_flush(LDS &L, uint32_t &n_dup, const ScanArgs &a, const WgTables &T, UpdQueue &qu)
{
    const uint32_t lane = lane_id(), grp = lane >> 4, sub = lane & 15u;
    uint32_t nupd = 0;
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < n_dup; c0 += 4u) {
        const uint32_t c = c0 + grp;
        const uint32_t len = c < n_dup ? L.g_len[c] : 0u;
        const uint32_t cc = c < n_dup ? c : 0u;
        nupd += group_score<EBWT, 16>(a, T, qu, L.g_doc[cc][sub], EBWT ? L.g_sym[cc][sub] : 0u, len);
    }
    n_dup = 0;
    return nupd;
}

// lanes with `on` hand over the cluster staged at L.da / L.fl[p .. p+len); returns the table updates of
// the flushes it had to make (per lane, like the scoring routines)
template <int EBWT, typename LDS>
__device__ __forceinline__ uint32_t dup_push(LDS &L, uint32_t &n_dup, const ScanArgs &a, const WgTables &T, UpdQueue &qu,
                                             bool on, uint32_t p, uint32_t len)
{
    uint64_t m = __ballot(on);
    uint32_t nupd = 0;
    while (m) {                                            // wave-uniform; one round unless the store fills up
        if (n_dup == LDS::NDUP) nupd += dup_flush<EBWT>(L, n_dup, a, T, qu);
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
        const bool take = on && rank < LDS::NDUP - n_dup;
        if (take) {
            const uint32_t slot = n_dup + rank;
            for (uint32_t k = 0; k < len; ++k) { L.g_doc[slot][k] = L.da[p + k]; if (EBWT) L.g_sym[slot][k] = L.fl[p + k]; }
            L.g_len[slot] = (uint8_t)len;
        }
        const uint64_t mt = __ballot(take);
        n_dup += (uint32_t)__popcll(mt);
        m &= ~mt; on = on && !take;
    }
    return nupd;
}

// ---- cluster scoring ----------------------------------------------------------------------
// One lane per cluster of the list (<= SMALL_MAX symbols, staged at L.da/L.fl[s..s+len)).  The
// cluster's read and genome positions are bit masks cut out of the window's mask bytes; the
// lane walks reads x genomes by popping bits (typically one or two pairs).  With all documents
// distinct -- checked only among documents of the same kind, the only ones that can repeat --
// a pair scores iupac_match of its two symbols (EBWT) or 1; a cluster with a repeated document
// goes to the general list instead.
template <int EBWT, typename LDS>
__device__ __forceinline__ uint32_t cluster_pairs(LDS &L, UpdQueue &qu, const ScanArgs &a, const uint16_t *list,
                                                  uint32_t n, uint32_t k_base, uint32_t &nD)
{
    const uint32_t lane = lane_id();
    const uint32_t k = k_base + lane;
    const bool on = k < n;
    const uint32_t item = list[on ? k : 0u];
    const uint32_t s = item & 0xFFFu, len = on ? (item >> 12) + 1u : 0u;
    const uint32_t kb = s >> 3, lenm = (1u << len) - 1u;
    const uint32_t rbits = (uint32_t)L.rb[kb] | ((uint32_t)L.rb[kb + 1u] << 8) | ((uint32_t)L.rb[kb + 2u] << 16);
    const uint32_t rmask = (rbits >> (s & 7u)) & lenm, gmask = ~rmask & lenm;
    // repeated document?  (reads among reads, genomes among genomes)
    uint32_t dup = 0;
#pragma unroll 1
    for (int kind = 0; kind < 2; ++kind) {
        uint32_t mi = kind ? gmask : rmask;
        while (__ballot((mi & (mi - 1u)) != 0u)) {
            const uint32_t rest = mi & (mi - 1u);
            const uint32_t di = L.da[s + (rest ? (uint32_t)__builtin_ctz(mi) : 0u)];
            uint32_t mj = rest;
            while (__ballot(mj != 0u)) {
                const bool actj = mj != 0u;
                const uint32_t dj = L.da[s + (actj ? (uint32_t)__builtin_ctz(mj) : 0u)];
                dup |= (uint32_t)(actj && dj == di);
                mj &= mj - 1u;
            }
            mi = rest;
        }
    }
    const uint64_t dm = __ballot(dup != 0u);
    if (dup) L.listD[nD + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = (uint16_t)item;
    nD += (uint32_t)__popcll(dm);
    uint32_t nupd = 0;
    uint32_t mr = dup ? 0u : rmask;
    while (__ballot(mr != 0u)) {
        const bool actr = mr != 0u;
        const uint32_t i = actr ? (uint32_t)__builtin_ctz(mr) : 0u;
        mr &= mr - 1u;
        const uint32_t rdoc = L.da[s + i], rsym = L.fl[s + i] & F_SYM;
        uint32_t mg = actr ? gmask : 0u;
        while (__ballot(mg != 0u)) {
            const bool actg = mg != 0u;
            const uint32_t j = actg ? (uint32_t)__builtin_ctz(mg) : 0u;
            mg &= mg - 1u;
            const uint32_t gdoc = L.da[s + j];
            const uint32_t t = EBWT ? iupac_match(rsym, L.fl[s + j] & F_SYM) : 1u;
            nupd += emit(qu, a, actg && t, rdoc, gdoc, 1u);
        }
    }
    return nupd;
}

// General routine, one lane per cluster [s, s+len) staged in LDS (len <= SMALL_MAX), any mix of
// repeated documents.  For every read (first occurrence) and every genome (first occurrence)
// the counts / 16-bin histograms are rebuilt by walking the cluster.  Quadratic, rare.
template <int EBWT>
__device__ __forceinline__ uint32_t cluster_general(const uint32_t *da, const uint8_t *fl, UpdQueue &qu, const ScanArgs &a, bool on, uint32_t s, uint32_t len)
{
    const uint32_t e = on ? s + len : s;
    uint32_t nupd = 0;
    for (uint32_t p = s; __ballot(p < e); ++p) {
        const bool pon = p < e;
        const uint32_t fp = pon ? fl[p] : 0u;
        const bool isr = pon && (fp & F_READ);
        const uint32_t rdoc = isr ? da[p] : 0u;
        uint32_t earlier = 0, rcount = 0;
        uint32_t cr[4] = {0u, 0u, 0u, 0u};
        if (isr)
            for (uint32_t q = s; q < e; ++q) {
                const uint32_t same = (da[q] == rdoc);
                rcount += same;
                earlier |= same & (uint32_t)(q < p);
                if (EBWT) hist_add(cr, fl[q] & F_SYM, same);
            }
        const bool rlead = isr && !earlier;
        for (uint32_t q = s; __ballot(rlead && q < e); ++q) {
            const bool qon = rlead && q < e;
            const uint32_t fq = qon ? fl[q] : 0u;
            const bool isg = qon && (fq & F_GEN);
            const uint32_t gdoc = isg ? da[q] : 0u;
            uint32_t gearlier = 0, gcount = 0;
            uint32_t cg[4] = {0u, 0u, 0u, 0u};
            if (isg)
                for (uint32_t x = s; x < e; ++x) {
                    const uint32_t same = (da[x] == gdoc);
                    gcount += same;
                    gearlier |= same & (uint32_t)(x < q);
                    if (EBWT) hist_add(cg, fl[x] & F_SYM, same);
                }
            uint32_t t = 0;
            if (isg && !gearlier) t = EBWT ? pair_score(cr, cg) : (rcount < gcount ? rcount : gcount);
            nupd += emit(qu, a, t != 0u, rdoc, gdoc, t);          // counts <= SMALL_MAX: no wrap, no saturation
        }
    }
    return nupd;
}

// all clusters filed in the wave's list; returns the number of table cells incremented (per lane)
template <int EBWT, typename LDS>
__device__ __forceinline__ uint32_t score_lists(LDS &L, UpdQueue &qu, const ScanArgs &a, uint32_t nA)
{
    uint32_t nupd = 0, nD = 0;
    for (uint32_t k0 = 0; k0 < nA; k0 += 64u) nupd += cluster_pairs<EBWT>(L, qu, a, L.listA, nA, k0, nD);
    for (uint32_t k0 = 0; k0 < nD; k0 += 64u) {
        const uint32_t k = k0 + lane_id();
        const uint32_t item = L.listD[k < nD ? k : 0u];
        nupd += cluster_general<EBWT>(L.da, L.fl, qu, a, k < nD, item & 0xFFFu, (item >> 12) + 1u);
    }
    drain(qu, a);
    return nupd;
}

// Clusters of exactly 2 symbols, one lane per cluster: an accepted one holds one read and one genome (never a repeated document), so it is ONE
// pair -- two documents, the roles by one compare, one compatibility lookup, at most one queue entry per lane (slots from a ballot, no prefix sum).
// Real collections are mostly such clusters (98 % in the text-derived fixture, ~245 per window): a dense window lists them apart from the others
// when it builds its cluster list (round 6: the heads of 2-symbol clusters are bit arithmetic on the chunk's head mask), so a round of them reads
// 64 positions and scores them -- no length from the staged head bits, no filing of the other clusters in between.
// (Rounds 3-5 found them inside the rounds over ALL clusters: list entry, 32 head bits, length, classification, then this routine with its read
// bit from the staged mask: ~140 vector instructions and 11 LDS instructions a round of 64, five rounds a window of text -- half that scan.
// Scoring them where they lie instead -- mask word by mask word, position 64 j + l on lane l, documents by a conflict-free ds_read2 -- was built
// and measured in round 6 and is SLOWER: 16 words a window at 24 % of the lanes cost more LDS and vector instructions than 4 full rounds: text
// scan 0.276 -> 0.285-0.307 ms, profiles/r06_inplace_ab.txt.)
template <int EBWT, typename LDS>
__device__ __forceinline__ uint32_t score_len2(LDS &L, const WgTables &T, UpdQueue &qu, const ScanArgs &a, bool on, uint32_t p)
{
    const uint32_t d0 = L.da[p], d1 = L.da[p + 1u];
    const bool r0 = d0 < a.n_reads;                                          // position p is the read; else p + 1 is
    bool hit = on;
    if (EBWT) hit = hit && ((T.compatb[L.fl[p]] >> T.symidx[L.fl[p + 1u]]) & 1u);
    const uint32_t rd = r0 ? d0 : d1, gd = (r0 ? d1 : d0) - a.n_reads;
    if (qu.direct) { put_rec(qu, a, hit, rd, gd); return hit ? 1u : 0u; }
    const uint64_t m = EBWT ? __ballot(hit) : __ballot(on);
    const uint32_t tot = (uint32_t)__popcll(m);
    while (qu.n + tot > qu.cap) drain(qu, a);
    if (hit) {
        const uint32_t slot = qu.n + rank_in(m);
        qu.qr[slot] = rd; qu.qg[slot] = gd | (1u << T_SHIFT);               // (document ids are range-checked by the drains, on full waves)
    }
    qu.n += tot;
    return hit ? 1u : 0u;
}

// =========================================================================================
// Round 3 back end of the scan: the same results with about a third of the vector instructions.
// (PMC, configs[2], per 1024-position window, round 2: staging 131, chunk acceptance 82, cluster list 159,
// 2-4-symbol round 214, rows 212 vector instructions; the six predicated emission blocks of the 2-4-symbol routine alone
// compiled to 26 each.)
// =========================================================================================

// Clusters of 2..4 symbols, one lane per cluster.  The pairs that join a read with a genome come as a LIST from a
// table indexed by (read bits, length): at most four slots, so the emission is four short blocks with static queue
// offsets instead of six; the documents of a slot are re-read from the staged window by position (a per-lane
// register index would cost a select chain).  Documents are not range-checked here: the drains do that on full
// waves.  Two equal documents (only documents of one kind can be equal: six plain compares, positions past the
// cluster carry impossible ids) hand the cluster to the wave's repeat store.
template <int EBWT, typename LDS>
__device__ __forceinline__ uint32_t score_small3(LDS &L, const WgTables &T, UpdQueue &qu, uint32_t &n_dup, const ScanArgs &a,
                                                 bool on, uint32_t p, uint32_t len)
{
    const uint32_t rmask = bits_at(L.rb, p) & ((1u << len) - 1u);
    const uint32_t pl = T.pairs[rmask | (((len - 1u) & 3u) << 4)];          // len == 0 (lane off): an entry without pairs
    uint32_t d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = L.da[p + i];
    d[2] = len > 2u ? d[2] : 0xFFFFFFFEu;
    d[3] = len > 3u ? d[3] : 0xFFFFFFFFu;
    const bool dup = on && ((d[0] == d[1]) | (d[0] == d[2]) | (d[0] == d[3]) | (d[1] == d[2]) | (d[1] == d[3]) | (d[2] == d[3]));
    const uint32_t nflush = dup_push<EBWT>(L, n_dup, a, T, qu, dup, p, len);
    const uint32_t np = pl >> 28;
    uint32_t hits = (1u << np) - 1u;                                          // bit e: slot e of the list scores
    if (EBWT) {
        uint32_t sy[4], cs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t by = L.fl[p + i]; sy[i] = T.symidx[by]; cs[i] = T.compatb[by]; }
        uint32_t compat6 = 0;
        {
            int pi = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i + 1; j < 4; ++j, ++pi) compat6 |= ((cs[i] >> sy[j]) & 1u) << pi;
        }
        uint32_t h = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) h |= ((compat6 >> ((pl >> (7 * e + 4)) & 7u)) & 1u) << e;
        hits &= h;
    }
    if (dup || !on) hits = 0u;
    const uint32_t nh = (uint32_t)__popc(hits);
    if (qu.direct && qu.two()) {
        // two sub-regions: a lane's records may go either way -- slot by slot through put_rec (two ballots and ranks each)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool he = (hits >> e) & 1u;
            if (e >= 2 && !__ballot(he)) continue;                            // wave-uniform
            const uint32_t rp = (pl >> (7 * e)) & 3u, gp = (pl >> (7 * e + 2)) & 3u;
            put_rec(qu, a, he, L.da[p + rp], L.da[p + gp] - a.n_reads);
        }
        return nh + nflush;
    }
    const uint32_t incl_all = wave_incl_scan(nh), total_all = rl32(incl_all, 63);
    // a batch adds at most 4 entries per lane = 256; where the queue is shorter than that (the 16-wave EBWT=1 kernel) a batch
    // that cannot fit even an emptied queue goes in two halves, slots 0..1 then 2..3 of the pair lists (at most 128 each)
    const uint32_t halves = total_all + 2u * 15u > qu.cap ? 2u : 1u;          // wave-uniform (a binned drain leaves up to 2 x 15 records behind in its line buffers, not in the queue)
#pragma unroll 1
    for (uint32_t half = 0; half < halves; ++half) {
        const uint32_t hp = halves == 1u ? hits : (half ? hits & 12u : hits & 3u);
        const uint32_t nhp = (uint32_t)__popc(hp);
        const uint32_t incl = halves == 1u ? incl_all : wave_incl_scan(nhp), total = halves == 1u ? total_all : rl32(incl, 63);
        while (qu.n + total > qu.cap) drain(qu, a);
        const uint32_t slot0 = qu.n + incl - nhp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool he = (hp >> e) & 1u;
            if (e >= 2 && !__ballot(he)) continue;                            // wave-uniform: most rounds have no cluster with 3 or 4 pairs
            if (he) {
                const uint32_t rp = (pl >> (7 * e)) & 3u, gp = (pl >> (7 * e + 2)) & 3u;
                const uint32_t slot = (EBWT || halves != 1u) ? slot0 + (uint32_t)__popc(hp & ((1u << e) - 1u)) : slot0 + (uint32_t)e;
                if (qu.direct) {                                              // (one sub-region: the finished record)
                    const uint32_t gd = L.da[p + gp] - a.n_reads;
                    qu.bad |= (uint32_t)(gd >= a.n_refs);
                    qu.qr[slot] = L.da[p + rp] * a.n_refs + gd;
                } else {
                qu.qr[slot] = L.da[p + rp];
                qu.qg[slot] = (L.da[p + gp] - a.n_reads) | (1u << T_SHIFT);
                }
            }
        }
        qu.n += total;
    }
    return nh + nflush;
}

// Clusters of 5..SMALL_MAX symbols by ROWS in fixed lane groups: G lanes per cluster (G = 8 for 5..8 symbols, 16 for
// 9..16), lane i of a group = position i of its cluster -- no prefix sums, no flags in LDS, and all rows of a cluster
// sit in one pass.  A row compares its document with those of the positions after it (equal: the whole cluster goes to
// the repeat store -- the group learns it from one ballot) and, if it is of the cluster's MORE COMMON kind, emits its
// pairs with the elements of the rarer kind, before and after it: the emission loop runs min(reads, genomes) times
// (usually once) instead of once per partner of a lone read.  `list`: entries position | (len-1) << 12.
template <int EBWT, int G, typename LDS>
__device__ __forceinline__ uint32_t score_rows3(LDS &L, const WgTables &T, UpdQueue &qu, uint32_t &n_dup, const ScanArgs &a,
                                                const uint16_t *list, uint32_t n)
{
    constexpr uint32_t LG = G == 8 ? 3u : 4u, PER = 64u / (uint32_t)G;
    static_assert(G == 8 || G == 16, "groups of 8 or 16 lanes");
    const uint32_t lane = lane_id(), g = lane >> LG, i = lane & ((uint32_t)G - 1u);
    uint32_t nupd = 0;
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < n; c0 += PER) {
        const bool valid = c0 + g < n;
        const uint32_t it = valid ? list[c0 + g] : 0u;                          // one address per group: a broadcast read
        const uint32_t p = it & 0xFFFu, cl = valid ? (it >> 12) + 1u : 0u;
        const bool on = i < cl;
        const uint32_t q = p + i;
        const uint32_t di = L.da[q];
        const uint32_t clm = (1u << cl) - 1u;
        const uint32_t rbc = bits_at(L.rb, p) & clm;                            // the cluster's read bits
        uint32_t dupb = 0;
#pragma unroll
        for (int k = 1; k < G; ++k) dupb |= (uint32_t)(L.da[q + (uint32_t)k] == di) << k;
        const uint32_t rem = on ? cl - 1u - i : 0u;                             // positions after i in the cluster
        const uint64_t dm = __ballot(on && (dupb & ((2u << rem) - 2u)));
        const bool gdup = (uint32_t)(dm >> (lane & ~((uint32_t)G - 1u))) & ((1u << G) - 1u);    // a row of my group met its document again
        const uint32_t ri = (rbc >> i) & 1u, nr = (uint32_t)__popc(rbc);
        const bool reads_common = 2u * nr > cl;
        // rows of the more common kind emit; partners: the elements of the other kind
        uint32_t partners = (on && !gdup && (ri != 0u) == reads_common) ? (ri ? ~rbc & clm : rbc) : 0u;
        const uint32_t ci = EBWT ? T.compatb[L.fl[q]] : 0xFFFFu;
        while (__ballot(partners != 0u)) {
            const bool act = partners != 0u;
            const uint32_t k = act ? (uint32_t)__builtin_ctz(partners) : 0u;
            partners &= partners - 1u;
            const uint32_t dj = L.da[p + k];
            bool hit = act;
            if (EBWT) hit = act && ((ci >> T.symidx[L.fl[p + k]]) & 1u);
            if (qu.direct) { put_rec(qu, a, hit, ri ? di : dj, (ri ? dj : di) - a.n_reads); nupd += (uint32_t)hit; continue; }
            const uint64_t m = __ballot(hit);
            while (qu.n + 64u > qu.cap) drain(qu, a);
            if (hit) {
                const uint32_t slot = qu.n + rank_in(m);
                qu.qr[slot] = ri ? di : dj;
                qu.qg[slot] = ((ri ? dj : di) - a.n_reads) | (1u << T_SHIFT);
            }
            qu.n += (uint32_t)__popcll(m);
            nupd += (uint32_t)hit;
        }
        if (dm) nupd += dup_push<EBWT>(L, n_dup, a, T, qu, valid && gdup && i == 0u, p, cl);
    }
    return nupd;
}

// ---- window loads: lane l holds positions 64 j + l (j < PPL) of the window -- every load
// instruction reads 64 consecutive elements, and the wave ballot of a comparison on register j IS
// mask word j -- the ebwt bytes 256 k + 4 l .. + 3 (k < PPL/4), and position WIN + l of the
// read-ahead (lanes < HALO).  Elements at or beyond n_avail read as 0.
struct WinRegs { uint32_t lv[PPL], dv[PPL], bv[PPL / 4], hl, hd, hb; };
typedef uint32_t __attribute__((may_alias, aligned(1))) u32u;    // ebwt words: any byte alignment

// the three input arrays are read once: non-temporal loads keep them from displacing the score table
// (the target of the compare-and-swaps) in the caches
#ifdef LIME_PLAIN_LOADS
#define LIME_STREAM_LOAD(p) (*(p))
#else
#define LIME_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#endif

template <int EBWT>
__device__ __forceinline__ void window_load(WinRegs &t, const ScanArgs &a, uint64_t lo)
{
    const uint32_t lane = lane_id();
    const ScanArgs &c = EBWT ? cold(a) : a;                // EBWT = 1 (SGPRs short): array pointers and length as scalar loads per window, not held through the loop
    const uint32_t *lp = c.lcp + lo + lane, *dp = c.da + lo + lane;
    if (lo + WIN <= c.n_avail) {                           // wave-uniform: the whole window is data
#pragma unroll
        for (int j = 0; j < (int)PPL; ++j) { t.lv[j] = LIME_STREAM_LOAD(lp + 64 * j); t.dv[j] = LIME_STREAM_LOAD(dp + 64 * j); }
#pragma unroll
        for (int k = 0; k < (int)PPL / 4; ++k)
            t.bv[k] = EBWT ? LIME_STREAM_LOAD(reinterpret_cast<const u32u *>(c.ebwt + lo + 256u * (uint32_t)k + 4u * lane)) : 0u;
    } else {
        // the last window of the data: addresses clamped to the last element (what the padding
        // positions read is never used: the masks mark them), ebwt byte by byte
        const uint64_t left = c.n_avail > lo ? c.n_avail - lo : 0ull;      // valid positions of the window (> 0)
        const uint32_t last = (uint32_t)left - 1u;
#pragma unroll
        for (int j = 0; j < (int)PPL; ++j) {
            const uint32_t p = 64u * (uint32_t)j + lane, q = p < last ? p : last;
            t.lv[j] = c.lcp[lo + q]; t.dv[j] = c.da[lo + q];
            // opaque to the optimiser: otherwise it folds these loads with the fast path's into one load
            // through a selected 64-bit address per register (32 address pairs of VGPRs, no `nt`)
            asm volatile("" : "+v"(t.lv[j]), "+v"(t.dv[j]));
        }
#pragma unroll
        for (int k = 0; k < (int)PPL / 4; ++k) {
            uint32_t w = 0;
            if (EBWT)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t p = 256u * (uint32_t)k + 4u * lane + (uint32_t)b;
                    w |= (uint32_t)c.ebwt[lo + (p < last ? p : last)] << (8 * b);
                }
            asm volatile("" : "+v"(w));
            t.bv[k] = w;
        }
    }
    const uint64_t hp = lo + WIN + lane;
    const bool hok = lane < HALO && hp < c.n_avail;
    t.hl = hok ? c.lcp[hp] : 0u;
    t.hd = hok ? c.da[hp] : 0u;
    t.hb = (EBWT && hok) ? c.ebwt[hp] : 0u;
}

// =========================================================================================
// k_scan: the streaming scan.  Every wave is an independent worker over windows of WIN positions; a workgroup (one per
// CU) takes chunks of as many consecutive windows as it has waves, which draw them from an LDS counter; no workgroup
// barrier in the loop.
// MODE 0: detect + score; 1: count clusters per window and keep the window's masks for k_emit.
// Front end: lane-strided loads; mask words from wave ballots; lane l owns the CHUNK of positions
// [16 l, 16 l + 16): its head / read bits are 16-bit masks cut out of the mask words, cluster
// acceptance inside a chunk is a carry ripple on 16 bits (32-bit arithmetic), the segment of a
// chunk's LAST head is decided with wave ballots over "chunk has a head / a read / a genome" and
// one ds_bpermute, the read-ahead being one more chunk (wave-uniform).  Every lane then writes
// (position, length) of its accepted clusters into an LDS list at the slots a wave prefix sum gives
// it; the scoring rounds read that list 64 clusters at a time.
// =========================================================================================
template <int EBWT, int BIN>                 // EBWT == 0: no symbols are staged (a kilobyte less per wave); BIN: the line buffers of the record drains
struct alignas(16) ScanLdsT {
    uint32_t da[WPOS + SMALL_MAX];
    uint8_t fl[EBWT ? WPOS + SMALL_MAX : 16];
    alignas(8) uint8_t hb[WPOS / 8 + 14];    // head / read bit of every staged position (byte k = positions 8k..8k+7)
    alignas(8) uint8_t rb[WPOS / 8 + 14];
    // accepted clusters of the window, in order: start | min(len, 63) << 10.  The scoring rounds read it 64
    // entries at a time and write the list of the clusters of 5..SMALL_MAX symbols (start | (len-1) << 12) over
    // its already consumed head: the k-th such cluster is at most the k-th cluster read.
    uint16_t listM[WIN / 2];
    uint16_t m_tstart[64];                   // the round's clusters of 9..SMALL_MAX symbols (position | (len-1) << 12)
    static constexpr uint32_t QCAP = EBWT ? QCAP_SCAN_SHORT : QCAP_SCAN;
    uint32_t q_read[QCAP], q_gen[QCAP];
    uint32_t sub_n[MAX_SUB + 1];             // binned updates: records in each of the wave's sub-regions; [MAX_SUB]: offset of the wave's producer group in the workgroup's bin histogram
    uint32_t lfill[2], lbuf[BIN ? 2 * LBUF : 2];   // records waiting for their 64-byte line (drain_lines)
    uint32_t g_doc[EBWT ? DUP_SLOTS_E : DUP_SLOTS][SMALL_MAX];
    uint8_t g_sym[EBWT ? DUP_SLOTS_E : 1][SMALL_MAX], g_len[DUP_SLOTS];
    static constexpr uint32_t NDUP = EBWT ? DUP_SLOTS_E : DUP_SLOTS;
};

struct Ctx16 {
    uint32_t h, r, g;   // 16-bit masks of this lane's chunk
    uint32_t ah;        // heads of accepted, owned clusters that close inside window + read-ahead
    uint32_t e_suf;     // window position where the segment of the chunk's last head ends (NONE32: open)
    uint64_t HW, RW, GW;  // chunk has a head / a read / a genome (wave ballots)
    uint32_t info;      // fh | lh << 5 | pre_r << 10 | pre_g << 11 | suf_r << 12 | suf_g << 13
};

// An owned run that is still open where the shard's arrays end, with more of the collection behind them
// (ClusterLCP.cpp:246-264 would read on).  Longer than LIME_MAX_CLUSTER already: if it holds a read and a genome
// it is a cluster ClusterBWT_DA refuses (:558-562) -- error now; else whether it becomes one is decided by the
// shards after this one: leave its content in the shard's edge word for the host to combine (lime_combine_edges).
// Shorter (a caller with a halo below LIME_MAX_CLUSTER): cannot be decided here -> LIME_FLAG_HALO.
__device__ __forceinline__ void open_run_at_end(const ScanArgs &a, uint64_t len_so_far, bool has_r, bool has_g)
{
    if (len_so_far <= LIME_MAX_CLUSTER) { atomicOr(&cold(a).stats->flags, LIME_FLAG_HALO); return; }
    if (has_r && has_g) { atomicOr(&cold(a).stats->flags, LIME_FLAG_MAXLEN); return; }
    atomicOr(cold(a).edge, LIME_EDGE_OPEN | (has_r ? LIME_EDGE_OPEN_R : 0u) | (has_g ? LIME_EDGE_OPEN_G : 0u));
}

// H64 / R64 / G64: the read-ahead chunk (16 bits, wave-uniform).  own_lim <= WIN.
__device__ __forceinline__ Ctx16 chunk_context(uint32_t h, uint32_t r, uint32_t g, uint32_t H64, uint32_t R64, uint32_t G64,
                                                uint32_t own_lim)
{
    const uint32_t lane = lane_id();
    Ctx16 c;
    c.h = h; c.r = r; c.g = g;
    const bool has_h = h != 0u;
    c.HW = __ballot(has_h); c.RW = __ballot(r != 0u); c.GW = __ballot(g != 0u);
    const uint32_t fh = has_h ? (uint32_t)__builtin_ctz(h) : 16u;
    const uint32_t lh = has_h ? 31u - (uint32_t)__builtin_clz(h) : 0u;
    const uint32_t lowm = (1u << fh) - 1u;                                        // headless: the whole chunk
    const uint32_t pre_r = (r & lowm) != 0u, pre_g = (g & lowm) != 0u;
    const uint32_t him = has_h ? ((0xFFFFu << lh) & 0xFFFFu) : 0u;
    const uint32_t suf_r = (r & him) != 0u, suf_g = (g & him) != 0u;
    c.info = fh | (lh << 5) | (pre_r << 10) | (pre_g << 11) | (suf_r << 12) | (suf_g << 13);
    const uint64_t gt = (lane == 63u) ? 0ull : (~0ull << (lane + 1u));
    const uint64_t above = c.HW & gt;
    const bool next_lane = above != 0ull;
    const uint32_t wn = next_lane ? (uint32_t)__builtin_ctzll(above) : 64u;
    const uint64_t between = gt & (wn >= 64u ? ~0ull : ((1ull << wn) - 1ull));   // headless chunks after this one
    const uint32_t mid_r = (c.RW & between) != 0ull, mid_g = (c.GW & between) != 0ull;
    const uint32_t nx = (uint32_t)__shfl((int)c.info, next_lane ? (int)wn : (int)lane);
    // no head in a later chunk of the window: the read-ahead chunk comes next
    const uint32_t hfh = H64 ? (uint32_t)__builtin_ctz(H64) : 16u, hlow = (1u << hfh) - 1u;
    const uint32_t n_r = next_lane ? (nx >> 10) & 1u : (uint32_t)((R64 & hlow) != 0u);
    const uint32_t n_g = next_lane ? (nx >> 11) & 1u : (uint32_t)((G64 & hlow) != 0u);
    const bool has_next = next_lane || H64 != 0u;
    c.e_suf = next_lane ? PPL * wn + (nx & 31u) : (H64 ? WIN + hfh : NONE32);
    const uint32_t acc_suf = (has_next && (suf_r | mid_r | n_r) && (suf_g | mid_g | n_g)) ? 1u : 0u;
    // "segment contains a read / a genome" gathered onto the segment's head bit by a carry ripple on 16
    // bits: in bit-reversed order a head is the TOP of its segment, and adding the seeds (Y) to the "may
    // receive from below" mask (Mr) ripples a carry through each segment up to its head
    const uint32_t Hr = __builtin_bitreverse32(h) >> 16, Mr = ~(Hr << 1) & 0xFFFFu;
    uint32_t Xr = __builtin_bitreverse32(r) >> 16, Y = (Xr << 1) & Mr;
    const uint32_t RH = (Xr | (((Mr + Y) ^ Mr) & Mr) | Y) & Hr;
    Xr = __builtin_bitreverse32(g) >> 16; Y = (Xr << 1) & Mr;
    const uint32_t GH = (Xr | (((Mr + Y) ^ Mr) & Mr) | Y) & Hr;
    uint32_t ah = __builtin_bitreverse32(RH & GH) >> 16;
    ah = (ah & ~(1u << lh)) | (acc_suf << lh);                     // last head: decided with the chunks after
    const uint32_t c0 = PPL * lane;                                // ownership: window part owned by this shard
    ah &= own_lim >= c0 + PPL ? 0xFFFFu : (own_lim <= c0 ? 0u : ((1u << (own_lim - c0)) - 1u));
    c.ah = has_h ? ah : 0u;
    return c;
}

// BIN 1 (MODE 0 only): table updates leave the kernel as records (binned update path) instead of compare-and-swaps.
// waves per SIMD the kernel is compiled for.  (The binned EBWT=0 scan was tried at four -- 128 VGPRs, 39.6 KB of LDS per
// workgroup with a 1024-bin histogram, no spills: same time as at three, 2.08 vs 2.10 ms on configs[2]; the scan is
// bound by the vector ALU's issue rate and the memory system together, not by latency a fourth wave would hide.)

template <int EBWT, int MODE, int BIN>
__global__ __launch_bounds__((ScanCfg<EBWT, BIN>::wg)) __attribute__((amdgpu_waves_per_eu((ScanCfg<EBWT, BIN>::waves), (ScanCfg<EBWT, BIN>::waves)))) void k_scan(ScanArgs a)
{
    constexpr int SCANK_WG = ScanCfg<EBWT, BIN>::wg;
    static_assert(BIN == 0 || MODE == 0, "records are made by the scoring scan only");
    static_assert(BIN >= 0 && BIN <= 2, "0: compare-and-swap; 1: records through the update queue (any number of sub-regions); 2: records written by the scorers (one or two sub-regions)");
    typedef ScanLdsT<EBWT, BIN> ScanLds;
    __shared__ ScanLds lds[SCANK_WG / 64];
    __shared__ WgTables T;
    // per wave the in-flight compare-and-swap slots (entry read, genome | t, expected word: 3 x 64 NJ words, NJ slots per
    // lane: 4, or 2 in the 8-wave workgroup); in binned mode the workgroup's histogram of update records per table bin instead
    constexpr uint32_t NJ = SCANK_WG > 256 ? 2u : 4u;
    constexpr uint32_t FS = BIN ? BIN_MAX : (SCANK_WG / 64) * 192u * NJ;
    __shared__ uint32_t fslots[FS];
    __shared__ uint32_t wg_done, wg_next, wg_exit, wg_max_len, wg_rec_max;
    __shared__ unsigned long long wg_n_clusters, wg_n_updates;
    __shared__ uint64_t wg_slot[16];
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: LDS bases stay scalar
    ScanLds &L = lds[wave];
    constexpr bool binned = BIN != 0;
    if (binned) {
        for (uint32_t i = threadIdx.x; i < FS; i += SCANK_WG) fslots[i] = 0u;
        if (threadIdx.x == 0) wg_done = 0u;
    }
    if (threadIdx.x == 0) { wg_next = 0u; wg_exit = 0u; wg_max_len = 0u; wg_rec_max = 0u; wg_n_clusters = 0ull; wg_n_updates = 0ull; }
    if (threadIdx.x < 16) wg_slot[threadIdx.x] = 0ull;
    tables_init(T);                                        // the only workgroup barrier of the kernel
    const uint32_t n_win = a.n_tiles;
    constexpr uint32_t WPW = SCANK_WG / 64;
    UpdQueue qu; qu.qr = L.q_read; qu.qg = L.q_gen; qu.n = 0; qu.cap = ScanLds::QCAP;
    qu.async = true;
    if (!binned) { qu.nj = NJ; qu.fr = fslots + 192u * NJ * wave; qu.fg = qu.fr + 64u * NJ; qu.fe = qu.fr + 128u * NJ; }
    const uint32_t wave_gid = blockIdx.x * (SCANK_WG / 64) + wave;
    qu.binned = binned;
    qu.direct = BIN == 2;
    if (BIN == 2) {
        qu.sub_rb = cold(a).sub_rb; qu.sub_gb = cold(a).sub_gb;             // (sub_rb = ~0 for one sub-region: fused_dev_impl)
    }
    if (binned) {
        // the workgroup's waves count their records in groups of prod_waves: every group is one "producer" of k_part (more, smaller
        // producers = more partition workgroups per CU); the groups' histograms lie one after the other
        qu.out = cold(a).pool + (size_t)wave_gid * cold(a).n_sub * cold(a).cap_w; qu.hist = fslots; qu.sub_n = (lds_vu32 *)L.sub_n;
        qu.lbuf = L.lbuf; qu.lfill = (lds_vu32 *)L.lfill;
        if (lane < MAX_SUB) L.sub_n[lane] = 0u;
        if (lane < 2u) L.lfill[lane] = 0u;
        if (lane == 0u) L.sub_n[MAX_SUB] = (wave / cold(a).prod_waves) * cold(a).n_bins;   // where this wave's group counts (read back per drain: held in a register it costs the loop an SGPR)
    }
    // binned mode, end of a wave: its record count; the workgroup's last wave writes the bin histogram
    auto finish_binned = [&]() {
        const uint32_t n_sub = cold(a).n_sub, cap_w = cold(a).cap_w;
        if (lane < n_sub) {
            const uint32_t n = BIN == 2 ? (lane ? qu.base1 : qu.base0) : qu.sub_n[lane];
            cold(a).wave_cnt[(size_t)(blockIdx.x * (SCANK_WG / 64) + wave) * n_sub + lane] = n < cap_w ? n : cap_w;
            atomicMax(&wg_rec_max, n);
            if (n > cap_w) {                                 // the pass's first overflow also counts the pass as unsettled
                const uint32_t old = atomicOr(&cold(a).stats->flags, LIME_FLAG_POOL_FULL);
                if (!(old & LIME_FLAG_POOL_FULL)) atomicAdd(cold(a).sticky, 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(&wg_done, 1u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        if (old == SCANK_WG / 64 - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t nb = cold(a).n_bins, H = (uint32_t)(SCANK_WG / 64) / cold(a).prod_waves, np = gridDim.x * H;
            for (uint32_t i = lane; i < nb * H; i += 64u) {
                const uint32_t h = i / nb, b = i - h * nb;
                cold(a).counts[(size_t)b * np + blockIdx.x * H + h] = fslots[i];
            }
        }
    };
    // The workgroup takes chunks of its wave count of consecutive windows and its waves draw the windows of those chunks
    // one by one from an LDS counter: the waves of a CU do not run equally fast (with every wave on a fixed share,
    // configs[2], the older wave of each SIMD's pairs ended at 0.70 .. 0.78 of the kernel's time and the tail ran at a
    // fraction of the occupancy); drawn this way they end within a window of each other.  The chunks of the first
    // n_static rounds go round-robin over the workgroups; the XCDs do not run equally fast either (the odd ones ended 5 ..
    // 9 % later), so the chunks of the last rounds come from a device-wide counter, one atomic per chunk: the wave that
    // draws the first window of chunk k fetches chunk k + 1 and leaves it, tagged k + 2, in the LDS slot the others poll.
    auto take = [&]() -> uint32_t {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&wg_next, 1u);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        const uint32_t ksw = cold(a).n_static, ks = ksw & 0xFFFFFFu, ps = ksw >> 24;   // rounds of round-robin chunks | probe shift << 24
        const uint32_t k = i / WPW, j = i % WPW;
        uint32_t chunk = k * gridDim.x + blockIdx.x;
        if (k >= ks) {
            // slot k & 15 carries the tags k + 1 - 16 (or 0), k + 1, k + 1 + 16, ... in turn.  A wave that finds a LATER tag than its own was
            // parked, between its draw and this read, while its workgroup went through 16 more chunks: its chunk's number is gone.
            // Practically out of reach (every other wave of the workgroup would have to draw 16 windows meanwhile), but a spin
            // without an exit is not: such a wave -- or one that has polled for seconds -- flags the pass as failed and stops.
            uint64_t v;
            uint32_t spins = 0;
            for (;;) {
                v = __hip_atomic_load(&wg_slot[k & 15u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t tag = (uint32_t)(v >> 32);
                if (tag == k + 1u) break;
                if (tag > k + 1u || ++spins > (1u << 24)) {
                    if (lane == 0) atomicOr(&cold(a).stats->flags, LIME_FLAG_INTERNAL);
                    return NONE32;                     // >= n_win: the wave ends
                }
                __builtin_amdgcn_s_sleep(2);
            }
            chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        }
        // (after chunk k is known: the workgroup's fetches follow each other, so its chunks ascend and a wave that draws a
        // window past the end can stop -- no chunk fetched later is still inside)
        if (j == 0u && k + 1u >= ks) {
            if (lane == 0) {
                const uint32_t g = atomicAdd(cold(a).dyn, 1u);
                __hip_atomic_store(&wg_slot[(k + 1u) & 15u], ((uint64_t)(k + 2u) << 32) | (uint64_t)(ks * gridDim.x + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return (chunk << ps) * WPW + j;                // (ps != 0: the density probe scans every 2^ps-th chunk only)
    };
    // end of a wave.  The counters of the pass are summed per workgroup in LDS and its last wave adds them to the device's:
    // the waves end together, and a few atomics per wave on the same few words took the last 0.1 .. 0.2 ms of the kernel
    // (configs[2]: 16 k atomics on one cache line).  The last wave of the last workgroup leaves the chunk counters at 0
    // for the next launch.
    auto wave_exit = [&](uint32_t tn, uint32_t tm, uint32_t tu) {
        if (lane == 0) {
            if (tn) atomicAdd(&wg_n_clusters, (unsigned long long)tn);
            if (tm) atomicMax(&wg_max_len, tm);
            if (tu) atomicAdd(&wg_n_updates, (unsigned long long)tu);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (atomicAdd(&wg_exit, 1u) == WPW - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                DevStats *st = cold(a).stats;
                const unsigned long long nc = __hip_atomic_load(&wg_n_clusters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), nu = __hip_atomic_load(&wg_n_updates, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t ml = __hip_atomic_load(&wg_max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), rm = __hip_atomic_load(&wg_rec_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (nc) atomicAdd(&st->n_clusters, nc);
                if (ml) atomicMax(&st->max_len, (unsigned long long)ml);
                if (MODE == 0 && nu) atomicAdd(&st->n_updates, nu);
                if (rm) atomicMax(&st->wave_records_max, rm);
                uint32_t *dyn = cold(a).dyn;
                if (atomicAdd(dyn + 1, 1u) == gridDim.x - 1u) {
                    __hip_atomic_store(dyn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(dyn + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    uint32_t win = take();
    if (win >= n_win) { if (binned) finish_binned(); wave_exit(0u, 0u, 0u); return; }
    WinRegs regs;
    window_load<EBWT>(regs, a, (uint64_t)win * WIN);
    uint32_t acc_n = 0, acc_max = 0, acc_upd = 0;          // per-lane partial counters, reduced once at the end
    uint32_t n_dup = 0;                                    // clusters waiting in the wave's dup store
    PT_DECL
#ifdef LIME_WALL_TIMING
    const uint64_t pt_wall0 = wall_clock64();
#endif
    for (;;) {
        PT_WAITVM PT(0)
#ifdef LIME_WALL_TIMING
        if (lane == 0 && win < (1u << 20)) atomicAdd(&g_winmark[win], ((wave_gid + 1u) & 0xFFFFu) | 0x10000u);
#endif
        const uint64_t lo = (uint64_t)win * WIN;
        const uint64_t n_own_ = (EBWT ? cold(a) : a).n_own;
        const uint32_t own_lim = (uint32_t)(n_own_ > lo ? (n_own_ - lo < WIN ? n_own_ - lo : (uint64_t)WIN) : 0ull);
        const uint64_t lim64 = (EBWT ? cold(a) : a).n_avail - lo;             // valid positions of the window + read-ahead: [0, lim)
        const uint32_t lim = (uint32_t)(lim64 < WPOS ? lim64 : (uint64_t)WPOS);
        // ---- stage the window in LDS and build the masks.  Loads are lane-strided (register j of lane l =
        // position 64 j + l: every load instruction reads 64 consecutive elements -- a lane reading its own 16
        // positions would touch 64 cache lines per instruction), so mask word j is the wave ballot of the
        // comparison on register j; v_writelane puts word j into lane j, and lane l then takes ITS 16 bits
        // (word l / 4, bits 16 (l % 4) ..) with one 64-bit permute per mask --------------------------------
        uint32_t hb, rb, gb, vb, H64, R64, G64, Hd64;
        {
            uint32_t hlo = 0, hhi = 0, rlo = 0, rhi = 0;
            static_assert(PPL == 16 && WIN / 64 == 16, "mask words are written lane by lane below");
#define LIME_WORD(J) { const uint64_t bh = __ballot(regs.lv[J] < a.alpha), br = __ballot(regs.dv[J] < a.n_reads); \
                       write_lane4<J>(hlo, hhi, rlo, rhi, (uint32_t)bh, (uint32_t)(bh >> 32), (uint32_t)br, (uint32_t)(br >> 32)); }
            LIME_WORD(0) LIME_WORD(1) LIME_WORD(2) LIME_WORD(3) LIME_WORD(4) LIME_WORD(5) LIME_WORD(6) LIME_WORD(7)
            LIME_WORD(8) LIME_WORD(9) LIME_WORD(10) LIME_WORD(11) LIME_WORD(12) LIME_WORD(13) LIME_WORD(14) LIME_WORD(15)
#undef LIME_WORD
            const uint32_t v16 = lim > WIN ? ((1u << (lim - WIN)) - 1u) & 0xFFFFu : 0u;       // data positions of the read-ahead
            Hd64 = (uint32_t)__ballot(lane < HALO && regs.hl < a.alpha) & v16;
            R64 = (uint32_t)__ballot(lane < HALO && regs.hd < a.n_reads) & v16;
            H64 = (Hd64 | ~v16) & 0xFFFFu; G64 = v16 & ~R64;
            hlo = write_lane<16>(hlo, H64); rlo = write_lane<16>(rlo, R64);
            uint64_t h = ((uint64_t)hhi << 32) | hlo, r = ((uint64_t)rhi << 32) | rlo;
            if (lim < WPOS) {                              // the end of the data: padding closes runs, is nobody's
                const uint32_t wl = 64u * lane;
                const uint64_t v = lim >= wl + 64u ? ~0ull : (lim <= wl ? 0ull : ((1ull << (lim - wl)) - 1ull));
                if (lane < NW) { h |= ~v; r &= v; }
            }
#pragma unroll
            for (int j = 0; j < (int)PPL; ++j) L.da[64 * j + (int)lane] = regs.dv[j];
            if (EBWT)
#pragma unroll
                for (int k = 0; k < (int)PPL / 4; ++k) reinterpret_cast<u32u *>(L.fl)[64 * k + (int)lane] = regs.bv[k];
            uint32_t lane_v = lane;                         // opaque per window: the lane masks below are compared here instead of living in SGPR pairs through the loop
            asm volatile("" : "+v"(lane_v));
            if (lane_v < HALO) { L.da[WIN + lane_v] = regs.hd; if (EBWT) L.fl[WIN + lane_v] = (uint8_t)regs.hb; }      // (lane_v: these addresses are recomputed per window, not held through the loop)
            if (lane_v <= WIN / 64) {
                *reinterpret_cast<u64a *>(&L.hb[8u * lane_v]) = h;
                *reinterpret_cast<u64a *>(&L.rb[8u * lane_v]) = r;
            }
            const uint32_t sh = 16u * (lane & 3u);
            hb = (uint32_t)(shfl64(h, (int)(lane >> 2)) >> sh) & 0xFFFFu;
            rb = (uint32_t)(shfl64(r, (int)(lane >> 2)) >> sh) & 0xFFFFu;
            const uint32_t c0_ = PPL * lane;
            vb = lim >= c0_ + PPL ? 0xFFFFu : (lim <= c0_ ? 0u : ((1u << (lim - c0_)) - 1u));
            gb = vb & ~rb;
        }
        const uint32_t c0 = PPL * lane;
        // ---- the table updates queued while the PREVIOUS window was scored leave now, before the next window's loads:
        // vmcnt counts stores and atomics with the loads, so a store issued after the loads keeps the wave waiting for
        // its acknowledgement when it wants to stage the loaded window (measured on configs[2]: 0.34 of 2.06 ms with the
        // stores issued from the scoring rounds).  Issued here they are older than the loads and long done by then.
        if (MODE == 0 && !ABL(8)) {
            if (ABL(13)) { qu.n = 0; qu.n1 = 0; } else
            if (BIN == 2) flush_direct(qu, a, false); else
            if (binned) drain_bin(qu, a); else drain(qu, a);
            asm volatile("" ::: "memory");                    // the loads below stay below
        }
        // ---- the next window's loads go out now and land while this one is processed ----------
        const uint32_t n_win_ = (EBWT ? cold(a) : a).n_tiles;
        const uint32_t next = take();
        if (next < n_win_) window_load<EBWT>(regs, a, (uint64_t)next * WIN);

        PT(1)
        if (!ABL(1)) {                               // LIME_ABLATE: timing experiments, cut after a phase
        const Ctx16 c = chunk_context(hb, rb, gb, H64, R64, G64, own_lim);
        // ---- the segment of the window's last head may still be open after the read-ahead (no head there): rare.
        // The scoring scan only notes it (start, what the window holds of it) for k_resolve_open, which walks the arrays
        // from the end of the window on; the detection pass keeps a summary per window for k_resolve<1> and k_emit.
        {
            if (MODE == 0) {
                if (H64 == 0u && c.HW != 0ull) {                                // wave-uniform
                    const uint32_t lw = 63u - (uint32_t)__clzll((long long)c.HW);
                    const uint32_t li = rl32(c.info, lw), last = PPL * lw + ((li >> 5) & 31u);
                    if (last < own_lim) {                                       // owned by this shard
                        const uint64_t whigh = (lw == 63u) ? 0ull : (~0ull << (lw + 1u));
                        const uint32_t suf = (((c.RW & whigh) || ((li >> 12) & 1u)) ? 1u : 0u) | (((c.GW & whigh) || ((li >> 13) & 1u)) ? 2u : 0u);
                        if (lane == 0) {
                            const uint32_t k = atomicAdd(&cold(a).stats->n_open, 1u);  // at most one per window: the list holds n_tiles records
                            OpenRec orec; orec.start = lo + last; orec.flags = suf; orec.pad = 0u;
                            cold(a).open[k] = orec;
                        }
                    }
                }
            } else {
            TileSummary sm;
            sm.first_head = NONE32; sm.last_head = NONE32;
            uint32_t pre = (c.RW ? 1u : 0u) | (c.GW ? 2u : 0u), suf = 0u;       // no head: the whole window is "prefix"
            if (c.HW) {                                                         // wave-uniform: scalar arithmetic
                const uint32_t fw = (uint32_t)__builtin_ctzll(c.HW), lw = 63u - (uint32_t)__clzll((long long)c.HW);
                const uint32_t fi = rl32(c.info, fw), li = rl32(c.info, lw);
                sm.first_head = PPL * fw + (fi & 31u); sm.last_head = PPL * lw + ((li >> 5) & 31u);
                const uint64_t wlow = (1ull << fw) - 1ull, whigh = (lw == 63u) ? 0ull : (~0ull << (lw + 1u));
                pre = (((c.RW & wlow) || ((fi >> 10) & 1u)) ? 1u : 0u) | (((c.GW & wlow) || ((fi >> 11) & 1u)) ? 2u : 0u);
                suf = (((c.RW & whigh) || ((li >> 12) & 1u)) ? 1u : 0u) | (((c.GW & whigh) || ((li >> 13) & 1u)) ? 2u : 0u);
            }
            if (lane == 0) { sm.pre = pre; sm.suf = suf; a.summ[win] = sm; }
            }
            // a run closed by padding instead of data while more data exists beyond the shard's
            // halo: the last data head in sight is owned and nothing but padding follows it
            if (lim < WPOS && Hd64 == 0u && !cold(a).eof) {
                const uint32_t dh = hb & vb;
                const uint64_t dw = __ballot(dh != 0u);
                if (dw) {
                    const uint32_t lw2 = 63u - (uint32_t)__clzll((long long)dw);
                    const uint32_t hl2 = rl32(dh, lw2);
                    const uint32_t bp = 31u - (uint32_t)__builtin_clz(hl2), sstar = PPL * lw2 + bp;
                    if (MODE == 1) {                           // detection alone has no length limit: the host starts over with one chunk
                        if (lane == 0 && sstar < own_lim) atomicOr(&cold(a).stats->flags, LIME_FLAG_HALO);
                    } else {
                        // what the open run holds so far: its head's chunk from the head on, the chunks after it, the read-ahead
                        const uint32_t cm = lane > lw2 ? 0xFFFFu : (lane == lw2 ? (0xFFFFu << bp) & 0xFFFFu : 0u);
                        const bool hr = __ballot((rb & vb & cm) != 0u) != 0ull || R64 != 0u;
                        const bool hg = __ballot((gb & cm) != 0u) != 0ull || G64 != 0u;
                        if (lane == 0 && sstar < own_lim) open_run_at_end(a, cold(a).n_avail - (lo + sstar), hr, hg);
                    }
                }
            }
        }
        PT(2)
        if (!ABL(3)) {
        // ---- the accepted clusters of the window: every lane lists the positions of its chunk's accepted heads at
        // the slots a wave prefix sum gives it (a short loop over its <= 8 head bits: six vector instructions a turn;
        // lengths are taken from the staged head bits later, by full waves) --------------------------------------
        const uint32_t cnt = (uint32_t)__popc(c.ah);
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t total = rl32(incl, 63);
        acc_n += cnt;
#ifdef LIME_DEBUG_CNT          // debug builds: the window's count of accepted clusters (tools/dbg_golden.py reads it back)
        if (MODE == 0 && lane == 0) a.tile_cnt[win] = total;
#endif
        if (MODE == 0) {
            // A dense window (more than 64 accepted clusters; real collections: hundreds, 98 % of them of 2 symbols) lists its 2-symbol clusters
            // apart: such a cluster is an accepted, owned head (c.ah) followed by a non-head and a head -- bits b + 1, b + 2 of the chunk's head
            // mask, the last two from the next chunk's (behind the last chunk: the read-ahead's).  The other clusters come first in the list,
            // the 2-symbol ones behind them: listM[total - total2 .. total).
            const bool defer = total > cold(a).dense_min;                       // (64; tests: 0 = every window)
            uint32_t m2c = 0, total2 = 0, k2 = 0;
            if (defer) {                                                        // wave-uniform
                uint32_t lv = lane;                                             // (opaque per window: the shuffle index is not held through the loop)
                asm volatile("" : "+v"(lv));
                const uint32_t nb = (uint32_t)__shfl((int)(c.h & 3u), (int)((lv + 1u) & 63u));
                const uint32_t h18 = c.h | ((lv == 63u ? H64 & 3u : nb) << 16);
                m2c = c.ah & ~(h18 >> 1) & (h18 >> 2);
                const uint32_t cnt2 = (uint32_t)__popc(m2c), incl2 = wave_incl_scan(cnt2);
                total2 = rl32(incl2, 63);
                k2 = (total - total2) + (incl2 - cnt2);
                acc_max = m2c && acc_max < 2u ? 2u : acc_max;
                {
                    uint32_t m = c.ah, k = (incl - cnt) - (incl2 - cnt2);
                    while (__ballot(m != 0u)) {
                        if (m) {
                            const uint32_t b = (uint32_t)__builtin_ctz(m);
                            const bool two = (m2c >> b) & 1u;
                            L.listM[two ? k2 : k] = (uint16_t)(c0 | b);
                            k2 += (uint32_t)two; k += (uint32_t)!two;
                            m &= m - 1u;
                        }
                    }
                }
            } else {
                uint32_t m = c.ah, k = incl - cnt;
                while (__ballot(m != 0u)) {
                    if (m) { L.listM[k++] = (uint16_t)(c0 | (uint32_t)__builtin_ctz(m)); m &= m - 1u; }
                }
            }
            PT(3)
            // Rounds of 64 clusters.  First the 2-symbol clusters of a dense window (listed apart: positions only, one pair each), then the others:
            // a round reads 64 positions, takes every cluster's length from the head bits (the next head after it), scores the short ones
            // (<= 4 symbols) at once and files those of 5..SMALL_MAX symbols (position | (len-1) << 12, over the already consumed head of the
            // list) for the rows routine.
            uint32_t nM = 0;
            const uint32_t n_rest = total - total2;
            if (!ABL(4) && !ABL(10))
#pragma unroll 1
            for (uint32_t base = 0; base < total2; base += 64u) {
                const uint32_t t = base + lane;
                const bool on = t < total2;
                const uint32_t p = on ? (uint32_t)L.listM[n_rest + t] & 0xFFFu : 0u;
                acc_upd += score_len2<EBWT>(L, T, qu, a, on, p);
            }
            if (!ABL(4) && !ABL(12))
#pragma unroll 1
            for (uint32_t base = 0; base < n_rest; base += 64u) {
                const uint32_t t = base + lane;
                const bool on = t < n_rest;
                const uint32_t item = on ? L.listM[t] : 0u;
                const uint32_t p = item & 0xFFFu;
                uint32_t len;
                {
                    const uint32_t w = bits_at(L.hb, p + 1u);             // head bits of p+1 .. p+32
                    len = w ? (uint32_t)__builtin_ctz(w) + 1u : 33u;
                    if (!on) len = 0u;
                    if (__ballot(len > SMALL_MAX)) {
                        if (len >= 33u) {                                 // rare: walk the head bytes to the end of the run
                            uint32_t q = p + 33u, e = WPOS;               // an accepted cluster closes before WPOS
                            while (q < WPOS) {
                                const uint32_t hbq = (uint32_t)L.hb[q >> 3] >> (q & 7u);
                                if (hbq) { e = q + (uint32_t)__builtin_ctz(hbq); break; }
                                q = (q | 7u) + 1u;
                            }
                            len = e - p;
                            if (len > MID_MAX) {                          // one workgroup per such cluster later
                                const uint32_t kk = atomicAdd(&cold(a).stats->n_big, 1u);
                                if (kk < cold(a).big_cap) { cold(a).big[kk].pStart = lo + p; cold(a).big[kk].len = len; }
                            }
                        }
                        // SMALL_MAX+1 .. MID_MAX symbols (rare): the whole wave is one lane group on the staged window, one cluster at a time
                        uint64_t mX = __ballot(len > SMALL_MAX && len <= MID_MAX);
                        while (mX) {
                            const uint32_t src = (uint32_t)__builtin_ctzll(mX);
                            mX &= mX - 1ull;
                            const uint32_t p0 = rl32(p, src), len0 = rl32(len, src);
                            const bool hvv = lane < len0;
                            acc_upd += group_score<EBWT, 64>(a, T, qu, hvv ? L.da[p0 + lane] : 0u, (EBWT && hvv) ? L.fl[p0 + lane] : 0u, len0);
                        }
                    }
                    acc_max = len > acc_max ? len : acc_max;
                }
                PT(4)
                if (ABL(10)) continue;
                const bool cM = len > 4u && len <= 8u;                     // rows in groups of 8 lanes, after the rounds
                const uint64_t mM = __ballot(cM);
                if (mM) {
                    if (cM) L.listM[nM + rank_in(mM)] = (uint16_t)(p | ((len - 1u) << 12));
                    nM += (uint32_t)__popcll(mM);
                }
                const bool cL = len > 8u && len <= SMALL_MAX;              // rare: groups of 16 lanes, at once (the list holds one round's worth)
                const uint64_t mL = __ballot(cL);
                const bool sm4 = len >= 2u && len <= 4u;
                acc_upd += score_small3<EBWT>(L, T, qu, n_dup, a, sm4, sm4 ? p : 0u, sm4 ? len : 0u);
                if (mL && !ABL(11)) {
                    if (cL) L.m_tstart[rank_in(mL)] = (uint16_t)(p | ((len - 1u) << 12));
                    acc_upd += score_rows3<EBWT, 16>(L, T, qu, n_dup, a, L.m_tstart, (uint32_t)__popcll(mL));
                }
                PT(5)
            }
            if (nM && !ABL(10) && !ABL(11) && !ABL(4)) acc_upd += score_rows3<EBWT, 8>(L, T, qu, n_dup, a, L.listM, nM);
            if (n_dup >= ScanLds::NDUP / 2u) acc_upd += dup_flush<EBWT>(L, n_dup, a, T, qu);
            PT(6)
        } else {
            // ---- count: records of the window, its masks for k_emit, the longest record -----------
            if (lane == 0) a.tile_cnt[win] = total;
            WinMasks &m = a.wmask[win];
            m.ah[lane] = (uint16_t)c.ah; m.h[lane] = (uint16_t)c.h; m.e_suf[lane] = c.e_suf;
            uint32_t mm = c.ah;
            while (mm) {
                const uint32_t b = (uint32_t)__builtin_ctz(mm);
                mm &= mm - 1u;
                const uint32_t p = c0 + b, hi = c.h >> (b + 1u);
                const uint32_t e = hi ? p + 1u + (uint32_t)__builtin_ctz(hi) : c.e_suf;
                acc_max = e - p > acc_max ? e - p : acc_max;
            }
        }
        }
        }
        if (next >= n_win_) break;
        win = next;
    }
    if (MODE == 0) {
        if (n_dup) acc_upd += dup_flush<EBWT>(L, n_dup, a, T, qu);
        if (BIN == 2) flush_direct(qu, a, true);                               // the last, partial lines too
        else {
        do drain(qu, a); while (qu.n != 0u || __ballot(qu.f_pend != 0u));      // until every update has landed
        if (binned) drain_bin(qu, a, true);                                    // the records still waiting for their line
        }
        if (binned) finish_binned();
    }
#ifdef LIME_PHASE_TIMING
    PT(7)
    if (MODE == 0 && lane == 0 && wave == 0 && blockIdx.x % 181u == 0u)
        printf("blk %u: wait %llu stage %llu ctx %llu list %llu book %llu small %llu medium %llu tail %llu\n", blockIdx.x,
               (unsigned long long)pt_acc[0], (unsigned long long)pt_acc[1], (unsigned long long)pt_acc[2], (unsigned long long)pt_acc[3],
               (unsigned long long)pt_acc[4], (unsigned long long)pt_acc[5], (unsigned long long)pt_acc[6], (unsigned long long)pt_acc[7]);
#endif
    wave_exit(wave_sum(acc_n), wave_max(acc_max), wave_sum(acc_upd));
#ifdef LIME_WALL_TIMING      // debug build: when each wave started and ended (tools/pt_wall.sh)
    if (MODE == 0 && lane == 0 && wave_gid < 8192u) { g_wall[2u * wave_gid] = pt_wall0; g_wall[2u * wave_gid + 1u] = wall_clock64(); }
#endif
}

// =========================================================================================
// k_emit: the (pStart, len) records of every window, in ascending pStart, from the masks the
// count pass kept (WinMasks) and the windows' record offsets.  A wave per window; lane l walks the
// accepted heads among positions [16 l, 16 l + 16).  The record of a run that crosses the window's
// read-ahead (k_resolve) goes last: its head is the window's last.
// =========================================================================================
__global__ __launch_bounds__(256) void k_emit(ScanArgs a)
{
    const uint32_t lane = lane_id();
    const uint32_t win = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (win >= a.n_tiles) return;
    const uint64_t lo = (uint64_t)win * WIN;
    const WinMasks &m = a.wmask[win];
    const uint32_t hh = m.h[lane], e_suf = m.e_suf[lane];
    uint32_t ahb = m.ah[lane];
    const uint32_t my_n = (uint32_t)__popc(ahb);
    const uint32_t x = wave_incl_scan(my_n);
    uint64_t at = a.tile_off[win] + x - my_n;
    while (ahb) {
        const uint32_t b = (uint32_t)__builtin_ctz(ahb);
        ahb &= ahb - 1u;
        const uint32_t p = PPL * lane + b, hi = hh >> (b + 1u);
        const uint32_t e = hi ? p + 1u + (uint32_t)__builtin_ctz(hi) : e_suf;
        lime_cluster_t rec; rec.pStart = a.pos_base + lo + p; rec.len = e - p;
        a.out[at++] = rec;
    }
    if (lane == 63u) {
        const CrossRec cr = a.cross[win];
        if (cr.len) {
            lime_cluster_t rec; rec.pStart = a.pos_base + cr.start; rec.len = cr.len;
            a.out[at] = rec;                              // lane 63: `at` is past all records of the window
        }
    }
}

// =========================================================================================
// k_resolve: closes the segment that is still open after a window's read-ahead, from the
// summaries of the windows after it (the reference's straddle loop, ClusterLCP.cpp:246-264,
// and EOF closure :244-245).  One thread per window.  Such a cluster is longer than the
// read-ahead, hence longer than SMALL_MAX.  MODE 0: push to the big list; 1: record for emit.
// =========================================================================================
template <int MODE>
__global__ __launch_bounds__(256) void k_resolve(ScanArgs a)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= a.n_tiles) return;
    if (MODE == 1) { CrossRec z; z.start = 0; z.len = 0; a.cross[t] = z; }
    if (MODE == 0 && t == 0u) {
        // what lies before the shard's first head belongs to a run of an earlier shard: its content for the
        // host's combination of the shards' edge words
        uint32_t lead = 0;
        for (uint32_t u = 0; u < a.n_tiles; ++u) {
            const TileSummary o = a.summ[u];
            lead |= ((o.pre & 1u) ? LIME_EDGE_LEAD_R : 0u) | ((o.pre & 2u) ? LIME_EDGE_LEAD_G : 0u);
            if (o.first_head != NONE32) { if ((uint64_t)u * WIN + o.first_head < a.n_avail) lead |= LIME_EDGE_LEAD_HEAD; break; }
        }
        atomicOr(a.edge, lead);
    }
    const TileSummary me = a.summ[t];
    if (me.last_head == NONE32) return;
    const uint64_t s = (uint64_t)t * WIN + me.last_head;
    if (s >= a.n_own) return;                                   // owned by the next shard / padding
    uint32_t fl = me.suf;
    uint64_t e = a.n_avail;
    bool closed_by_data = false;
    for (uint32_t u = t + 1u; u < a.n_tiles; ++u) {
        const TileSummary o = a.summ[u];
        fl |= o.pre;
        if (o.first_head != NONE32) { e = (uint64_t)u * WIN + o.first_head; closed_by_data = true; break; }
    }
    if (e >= a.n_avail) { e = a.n_avail; closed_by_data = false; }
    if (!closed_by_data && !a.eof) {                            // still open where the shard's arrays end
        if (MODE == 1) { if (e >= ((uint64_t)t + 1u) * WIN + HALO) atomicOr(&a.stats->flags, LIME_FLAG_HALO); }   // (inside the read-ahead: k_scan saw it)
        else open_run_at_end(a, e - s, (fl & 1u) != 0u, (fl & 2u) != 0u);
        return;
    }
    if (e < ((uint64_t)t + 1u) * WIN + HALO) return;            // seen (and handled) inside the window's read-ahead
    const uint64_t len = e - s;
    if (fl != 3u || len < 2u) return;
    atomicAdd(&a.stats->n_clusters, 1ull);
    atomicMax(&a.stats->max_len, (unsigned long long)len);
    atomicAdd(&a.stats->n_cross, 1u);
    if (MODE == 1) {
        CrossRec c; c.start = s; c.len = len; a.cross[t] = c;
        a.tile_cnt[t] += 1u;
    } else {
        if (len > LIME_MAX_CLUSTER) { atomicOr(&a.stats->flags, LIME_FLAG_MAXLEN); return; }
        const uint32_t k = atomicAdd(&a.stats->n_big, 1u);
        if (k < a.big_cap) { a.big[k].pStart = s; a.big[k].len = len; }
    }
}

// =========================================================================================
// k_resolve_open (scoring scan): a wave per noted segment -- the last head of a window whose run has no head in the
// window's read-ahead.  The wave walks lcp / da from the end of that window to the run's end (the reference's straddle
// loop, ClusterLCP.cpp:246-264; EOF closure :244-245) and decides like k_resolve.  Wave 0 first leaves in the shard's
// edge word what lies before the shard's first head (it belongs to a run of an earlier shard).
// =========================================================================================
__device__ __forceinline__ void resolve_open_body(const ScanArgs &a, uint32_t block, uint32_t n_blocks)
{
    const uint32_t lane = lane_id();
    const uint32_t wave = block * 4u + (threadIdx.x >> 6), n_waves = n_blocks * 4u;
    // positions [from, n_avail): where the first head is (n_avail: none) and what the positions before it hold
    auto walk = [&](uint64_t from, uint32_t &fl) -> uint64_t {
        for (uint64_t i = from; i < a.n_avail; i += 64u) {
            const uint64_t p = i + lane;
            const bool ok = p < a.n_avail;
            const bool head = ok && a.lcp[p] < a.alpha;
            const bool isr = ok && a.da[p] < a.n_reads;
            const uint64_t mh = __ballot(head);
            const uint64_t below = mh ? ((1ull << (uint32_t)__builtin_ctzll(mh)) - 1ull) : ~0ull;
            if (__ballot(ok && isr) & below) fl |= 1u;
            if (__ballot(ok && !isr) & below) fl |= 2u;
            if (mh) return i + (uint32_t)__builtin_ctzll(mh);
        }
        return a.n_avail;
    };
    if (wave == 0) {
        uint32_t fl = 0;
        const uint64_t h0 = walk(0, fl);
        if (lane == 0) atomicOr(a.edge, ((fl & 1u) ? LIME_EDGE_LEAD_R : 0u) | ((fl & 2u) ? LIME_EDGE_LEAD_G : 0u) | (h0 < a.n_avail ? LIME_EDGE_LEAD_HEAD : 0u));
    }
    const uint32_t n_open = a.stats->n_open < a.n_tiles ? a.stats->n_open : a.n_tiles;
    for (uint32_t k = wave; k < n_open; k += n_waves) {
        const uint64_t s = a.open[k].start;
        uint32_t fl = a.open[k].flags;
        const uint64_t e = walk((s / WIN + 1u) * WIN, fl);
        if (lane != 0) continue;
        if (e >= a.n_avail && !a.eof) { open_run_at_end(a, e - s, (fl & 1u) != 0u, (fl & 2u) != 0u); continue; }   // still open where the shard's arrays end
        const uint64_t len = e - s;
        if (fl != 3u || len < 2u) continue;
        atomicAdd(&a.stats->n_clusters, 1ull);
        atomicMax(&a.stats->max_len, (unsigned long long)len);
        atomicAdd(&a.stats->n_cross, 1u);
        if (len > LIME_MAX_CLUSTER) { atomicOr(&a.stats->flags, LIME_FLAG_MAXLEN); continue; }
        const uint32_t kb = atomicAdd(&a.stats->n_big, 1u);
        if (kb < a.big_cap) { a.big[kb].pStart = s; a.big[kb].len = len; }
    }
}
__global__ __launch_bounds__(256) void k_resolve_open(ScanArgs a) { resolve_open_body(a, blockIdx.x, gridDim.x); }

// =========================================================================================
// k_scan_tiles: exclusive prefix sum of the per-tile record counts (one workgroup).
// =========================================================================================
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t *cnt, uint64_t *off,
                                                     uint32_t n, unsigned long long *total)
{
    constexpr uint32_t PER = 16;                       // consecutive counters per thread and round
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u * PER) {
        const uint32_t i0 = base + tid * PER;
        uint32_t v[PER];
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { v[k] = (i0 + k < n) ? cnt[i0 + k] : 0u; mine += v[k]; }
        uint64_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint64_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        if (lane == 63u) wsum[wave] = x;
        __syncthreads();
        uint64_t wpre = 0;
        for (uint32_t k = 0; k < wave; ++k) wpre += wsum[k];
        const uint64_t c = carry;
        uint64_t run = c + wpre + x - mine;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { if (i0 + k < n) off[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (tid == 1023u) carry = run;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// =========================================================================================
// Binned table updates, after the scan (upd_mode 1).  The scan left, per producer workgroup p and table
// bin b, the number of update records p's waves wrote to the pool (counts[b][p]) and the records
// themselves in per-wave regions.  Reference site of the updates: ClusterBWT_DA.cpp:178-184, 243-248.
//
// k_bin_rowscan: one wave per bin: counts[b][.] becomes its exclusive prefix over the producers (where p's
// records of bin b start inside the bin) and totals[b] the bin's size; k_scan_tiles turns the totals into
// bin bases.
// =========================================================================================
__global__ __launch_bounds__(256) void k_bin_rowscan(uint32_t *counts, uint32_t *totals, uint32_t n_bins, uint32_t n_prod)
{
    const uint32_t lane = lane_id();
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_bins) return;
    uint32_t *row = counts + (size_t)b * n_prod;
    uint32_t run = 0;
    for (uint32_t p0 = 0; p0 < n_prod; p0 += 64u) {
        const uint32_t p = p0 + lane;
        const uint32_t v = p < n_prod ? row[p] : 0u;
        const uint32_t incl = wave_incl_scan(v);
        if (p < n_prod) row[p] = run + incl - v;
        run += rl32(incl, 63);
    }
    if (lane == 0) totals[b] = run;
}

// The binned pass's two small launches behind the scan (round 5; they were four: each is 5 us of a pass whose partition takes 19 .. 160 us on the
// 10^8-symbol workloads): k_rowscan_resolve = k_bin_rowscan in its first blocks + k_resolve_open in 64 more; k_bin_bases = the bins' bases (what
// k_scan_tiles did with the totals) and, for the second level by tiles, the tiles before each bin (k_tile_bases) by one workgroup.
constexpr uint32_t RESOLVE_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_rowscan_resolve(ScanArgs a, uint32_t *counts, uint32_t *totals, uint32_t n_bins, uint32_t n_prod)
{
    const uint32_t rb = (n_bins + 3u) / 4u;
    if (blockIdx.x >= rb) { resolve_open_body(a, blockIdx.x - rb, RESOLVE_BLOCKS); return; }
    const uint32_t lane = lane_id();
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_bins) return;
    uint32_t *row = counts + (size_t)b * n_prod;
    uint32_t run = 0;
    for (uint32_t p0 = 0; p0 < n_prod; p0 += 64u) {
        const uint32_t p = p0 + lane;
        const uint32_t v = p < n_prod ? row[p] : 0u;
        const uint32_t incl = wave_incl_scan(v);
        if (p < n_prod) row[p] = run + incl - v;
        run += rl32(incl, 63);
    }
    if (lane == 0) totals[b] = run;
}
__global__ __launch_bounds__(1024) void k_bin_bases(const uint32_t *totals, uint64_t *binbase, uint32_t *tbase, uint32_t n_bins, uint32_t tile)
{
    constexpr uint32_t PER = (BIN_MAX + 1023u) / 1024u;
    __shared__ uint64_t wsum[16];
    __shared__ uint32_t tsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, i0 = tid * PER;
    uint32_t v[PER], t[PER];
    uint64_t mine = 0; uint32_t minet = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { v[k] = i0 + k < n_bins ? totals[i0 + k] : 0u; t[k] = (v[k] + tile - 1u) / tile; mine += v[k]; minet += t[k]; }
    uint64_t x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
    const uint32_t xt = wave_incl_scan(minet);
    if (lane == 63u) { wsum[wave] = x; tsum[wave] = xt; }
    __syncthreads();
    uint64_t run = x - mine; uint32_t runt = xt - minet;
    for (uint32_t k = 0; k < wave; ++k) { run += wsum[k]; runt += tsum[k]; }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        if (i0 + k < n_bins) { binbase[i0 + k] = run; if (tbase) tbase[i0 + k] = runt; }
        run += v[k]; runt += t[k];
        if (i0 + k == n_bins - 1u) { binbase[n_bins] = run; if (tbase) tbase[n_bins] = runt; }
    }
}

// (bins as wide as a region: the bin bases are the region bases; wider bins go through k_part2 first)
//
// Both partition kernels move records TILE by TILE through LDS: a tile's records are ranked inside their bin with
// one returning LDS add each, an exclusive scan of the tile's bin counts gives every bin a run of LDS slots, the
// records go to their slots together with their final position, and the tile leaves LDS slot by slot -- so a wave's
// store instruction writes runs of consecutive positions instead of 64 scattered dwords (scattered 4-byte stores
// cost the CU's address path about 3 cycles per lane: 0.8 ms per 1.2e8 records and level, measured).
#ifndef LIME_PART_PER
#define LIME_PART_PER 16
#endif
constexpr int PART_WG = 512;
constexpr uint32_t PART_PER = LIME_PART_PER, PART_TILE = PART_WG * PART_PER;   // 8192 records per tile, 64 KB of (position, record)
constexpr uint32_t ROW_STRIDE = PART_TILE;                              // 16-bit records from one second-level tile row to the next (padding it -- 256 B, 4.25 KB -- changed nothing)

// exclusive prefix of cnt[0 .. nb) into toff[0 .. nb), nb <= PART_WG * 8; all threads of the workgroup call it
// (barriers inside: cnt is complete on entry, toff on exit)
template <int WG = PART_WG>
__device__ __forceinline__ void part_scan(const uint32_t *cnt, uint32_t *toff, uint32_t nb, uint32_t *wsum)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (nb + WG - 1u) / WG, b0 = tid * per;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) mine += b0 + k < nb ? cnt[b0 + k] : 0u;
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t k = 0; k < wave; ++k) run += wsum[k];
    for (uint32_t k = 0; k < per; ++k) if (b0 + k < nb) { toff[b0 + k] = run; run += cnt[b0 + k]; }
    __syncthreads();
}

// k_part: workgroup p moves the records of producer p (a group of prod_waves scan waves: their pool segments, one after the
// other) into their bins; a record leaves as 4 bytes: cell offset inside the bin | t << bin_shift.  Positions are 32-bit
// (the host keeps a pass below 2^32 records).
// Round 4 (the round-3 kernel issued 60 instructions per 64 records -- 36 vector, 15 scalar, 6 LDS, 2 memory -- and ran at 2.1
// cycles per record and CU whatever the number of bins or of workgroups per CU): (1) a tile is COUNTED per bin (LDS add, nothing
// returned), the counts are scanned, and each record then takes the next slot of its bin's cursor -- the order inside a bin is
// free -- so no rank travels in registers between the passes; (2) records are loaded four at a time (16-byte loads) and the
// tile leaves LDS four slots at a time: consecutive positions go out as ONE 16-byte store (the hardware takes them at any
// 4-byte alignment, tools/store_bench.hip), the others as single words; (3) the bins' global cursors live in the registers of
// the threads that scan them; the counters are cleared by the scan, and the NEXT tile is counted while this one is written
// out: three barriers a tile.
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));   // four words at any 4-byte alignment
#if defined(LIME_PART_TIMING) || defined(LIME_APPLY_TIMING) || defined(LIME_SORT_TIMING)      // debug builds: cycles of k_part's (k_apply_tiles') phases, summed over the first wave of every workgroup (tools/r04_part_phases.sh, tools/r04_apply_phases.sh)
__device__ unsigned long long g_part_pt[8];
extern "C" int lime_debug_part_times(unsigned long long *out)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_part_pt), sizeof(g_part_pt));
    void *p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_part_pt)); (void)hipMemset(p, 0, sizeof(g_part_pt));
    return rc;
}
#define PT_DECL_ uint64_t pp_t = __builtin_readcyclecounter(), pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PT_(i) { const uint64_t n_ = __builtin_readcyclecounter(); pp_acc[i] += n_ - pp_t; pp_t = n_; }
#ifndef LIME_PT_WAVE
#define LIME_PT_WAVE 0               // the wave of every workgroup whose cycles are summed (tools/r04_part_phases_waves.sh)
#endif
#define PT_END_ if (threadIdx.x == 64 * LIME_PT_WAVE) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_part_pt[i_], (unsigned long long)pp_acc[i_]); }
#endif
#ifdef LIME_PART_TIMING
#define PP_DECL PT_DECL_
#define PP(i) PT_(i)
#define PP_END PT_END_
#else
#define PP_DECL
#define PP(i)
#define PP_END
#endif
#ifdef LIME_SORT_TIMING
#define ST_DECL PT_DECL_
#define ST(i) PT_(i)
#define ST_END PT_END_
#define ST_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#define ST_DECL
#define ST(i)
#define ST_END
#define ST_WAITVM
#endif
#ifdef LIME_APPLY_TIMING
#define AP_DECL PT_DECL_
#define AP(i) PT_(i)
#define AP_END PT_END_
#define AP_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#define AP_DECL
#define AP(i)
#define AP_END
#define AP_WAITVM
#endif

// WGS threads and tiles of 16 WGS records: 512 / 8192, or -- few bins: the runs stay long enough -- 256 / 4096 with twice as many
// workgroups per CU: a tile is a chain of short phases between barriers, and what hides their latencies is other workgroups
// P64 (round 5): a pass whose record pool holds 2^32 records or more (N = 1e10 at the update density of real text: 2.4 .. 3.9e9 records) --
// positions in `out` are 64-bit: the bins' cursors are 64-bit registers, a bin's (position - slot) is a 64-bit word in LDS, and the high part
// of a slot's position travels through the stage in the record's free bits above t (t is 1 here: a score of t left the scan as t records),
// bits bin_shift + 1 .. 31: six bits at the widest bins, 2^38 records.  Rounds 1-4 sent such a pass to the compare-and-swap path.
template <int WGS, uint32_t NB_MAX, bool P64>
__global__ __launch_bounds__(WGS) void k_part(ScanArgs a, const uint64_t *binbase, uint32_t *out)
{
    constexpr uint32_t PART_WG = WGS, PART_TILE = WGS * PART_PER, PART_BPT = (NB_MAX + WGS - 1) / WGS;   // (shadow the file's constants)
    __shared__ uint4 stage4[PART_TILE / 2];                              // (position in out, record) per slot
    extern __shared__ __attribute__((aligned(8))) uint32_t part_lds[];   // per bin: tile count, cursor (LDS slot), position - slot (P64: two words)
    __shared__ uint32_t wsum[PART_WG / 64], tile_n_s;
    uint2 *stage = reinterpret_cast<uint2 *>(stage4);
    const uint32_t nb = a.n_bins, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t *cnt = part_lds, *cur = cnt + nb, *delta = cur + nb;
    u64a *delta64 = reinterpret_cast<u64a *>(part_lds + 2u * nb);        // (8-byte aligned: 2 nb words in front; takes the place of delta)
    typedef typename std::conditional<P64, uint64_t, uint32_t>::type pos_t;
    const uint32_t per = (nb + PART_WG - 1u) / PART_WG, b0 = tid * per;
    pos_t G[PART_BPT];                                                   // where this producer's records of bins b0 .. go next
#pragma unroll
    for (uint32_t k = 0; k < PART_BPT; ++k) {
        G[k] = 0u;
        if (k < per && b0 + k < nb) { G[k] = (pos_t)binbase[b0 + k] + a.counts[(size_t)(b0 + k) * gridDim.x + blockIdx.x]; cnt[b0 + k] = 0u; }
    }
    __syncthreads();
    const uint32_t sh = a.bin_shift, omask = (1u << sh) - 1u, tbit = 1u << sh;
    // the producer's segments -- (wave, sub-region): 32-bit records, the cell's high part is the sub-region's number -- as one
    // sequence of tiles
    const uint32_t n_seg = a.prod_waves * a.n_sub, seg0 = blockIdx.x * n_seg;
    // The producer's records -- its (wave, sub-region) pool segments one after the other, each padded to a multiple of four records (16-byte loads,
    // segments start on 64-byte lines) -- are ONE stream cut into tiles of PART_TILE (round 5).  Rounds 3-4 cut every segment into tiles of its own:
    // nothing lost where a segment holds many tiles, but on 10^8-symbol inputs a wave has 800 .. 6000 records and every producer walked 8 partly
    // filled tiles where 1 .. 6 full ones do (configs[1] binned: k_part 52 of the pass's 280 us; the text workload: 8 tiles of 0.72 instead of 5.8).
    // (one word per segment: its padded start | the segment's padding, 0 .. 3 records, in the two low bits -- a second array of counts was the 512 bytes
    // by which k_part_lines<true> at 477 bins no longer fitted a CU twice)
    __shared__ uint32_t segp_s[16u * MAX_SUB + 1u];
    for (uint32_t i = tid; i < n_seg; i += PART_WG) segp_s[i] = a.wave_cnt[seg0 + i];
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < n_seg; ++i) { const uint32_t n = segp_s[i]; segp_s[i] = run | ((0u - n) & 3u); run += (n + 3u) & ~3u; }
        segp_s[n_seg] = run;
    }
    __syncthreads();
    auto seg_p = [&](uint32_t i) { return segp_s[i] & ~3u; };
    auto seg_n = [&](uint32_t i) { const uint32_t w = segp_s[i]; return (segp_s[i + 1u] & ~3u) - (w & ~3u) - (w & 3u); };
    const uint32_t l_pad = segp_s[n_seg];                                // padded records of the producer
    // start in the stream, the segment that holds it; one: the tile's records all lie in that segment (the rule where segments are long: the tile
    // is then described by two wave-uniform words, tn records from the segment's offset v0 - start on, like rounds 3-4's tiles -- the per-group
    // meta words below cost the partition of N = 1e10 6 % when every tile used them)
    struct Tile { uint32_t v0, w0, tn, binoff; bool any, one; };
    auto tile_at = [&](uint32_t v0, uint32_t w0) {
        Tile t; t.v0 = v0; t.w0 = w0; t.any = v0 < l_pad; t.one = false; t.tn = 0u; t.binoff = 0u;
        if (t.any) {
            while (seg_p(t.w0 + 1u) <= v0) ++t.w0;
            const uint32_t end = v0 + PART_TILE < l_pad ? v0 + PART_TILE : l_pad;
            t.one = end <= seg_p(t.w0 + 1u);
            if (t.one) { const uint32_t left = seg_n(t.w0) - (v0 - seg_p(t.w0)); t.tn = left < PART_TILE ? left : PART_TILE; t.binoff = (t.w0 % a.n_sub) << (32u - sh); }
        }
        return t;
    };
    auto next_tile = [&](const Tile &c) { return tile_at(c.v0 + PART_TILE, c.w0); };
    // records 4 (j * PART_WG + tid) .. + 3 of the tile (16-byte loads); meta: per group of four how many of them are records (0 .. 4) and the
    // number of their sub-region (= high part of the cell), six bits a group
    auto load_tile = [&](const Tile &t, uint4 (&r)[PART_PER / 4], uint32_t &meta) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        meta = 0u;
        if (t.one) {                                                     // (groups past the tile's end read its last group again: never used, the passes look at tn)
            const u32x4 *src = reinterpret_cast<const u32x4 *>(a.pool + (size_t)(seg0 + t.w0) * a.cap_w + (t.v0 - seg_p(t.w0)));
            const uint32_t lastq = (t.tn - 1u) >> 2;
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) {
                const uint32_t q = j * PART_WG + tid;
                const u32x4 x = __builtin_nontemporal_load(src + (q < lastq ? q : lastq));
                r[j] = make_uint4(x.x, x.y, x.z, x.w);
            }
            return;
        }
#pragma unroll
        for (uint32_t j = 0; j < PART_PER / 4; ++j) {
            const uint32_t v = t.v0 + 4u * (j * PART_WG + tid);
            r[j] = make_uint4(0u, 0u, 0u, 0u);
            if (v < l_pad) {
                uint32_t w = t.w0;
                while (seg_p(w + 1u) <= v) ++w;
                const uint32_t off = v - seg_p(w), n = seg_n(w), vc = n - off < 4u ? n - off : 4u;
                const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.pool + (size_t)(seg0 + w) * a.cap_w + off));
                r[j] = make_uint4(x.x, x.y, x.z, x.w);
                meta |= (vc | ((w % a.n_sub) << 3)) << (6u * j);
            }
        }
    };
    auto count_tile = [&](const Tile &t, const uint4 (&r)[PART_PER / 4], uint32_t meta) {
        if (t.one) {
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) {
                const uint32_t i = 4u * (j * PART_WG + tid);
                const uint32_t v[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) if (i + k < t.tn) atomicAdd(&cnt[(v[k] >> sh) + t.binoff], 1u);
            }
            return;
        }
#pragma unroll
        for (uint32_t j = 0; j < PART_PER / 4; ++j) {
            const uint32_t vc = (meta >> (6u * j)) & 7u, bo = ((meta >> (6u * j + 3u)) & 7u) << (32u - sh);
            const uint32_t v[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) if (k < vc) atomicAdd(&cnt[(v[k] >> sh) + bo], 1u);
        }
    };
    Tile tc = tile_at(0u, 0u);
    if (!tc.any) return;
    uint4 rv[PART_PER / 4], v4[PART_PER / 4];
    uint32_t mv = 0, m4 = 0;                                             // the groups' meta words of rv / v4
    load_tile(tc, rv, mv);
#pragma unroll
    for (uint32_t j = 0; j < PART_PER / 4; ++j) v4[j] = rv[j];
    m4 = mv;
    count_tile(tc, v4, m4);
    Tile tn_ = next_tile(tc);
    if (tn_.any) load_tile(tn_, rv, mv);
    PP_DECL
    for (;;) {
        // ---- scan of the tile's counts: bin cursors (LDS slots), position - slot per bin; the counters go back to zero
        {
            uint32_t c[PART_BPT], mine = 0;
            PP(0)
            __syncthreads();                                             // the counts are complete
            PP(1)
#pragma unroll
            for (uint32_t k = 0; k < PART_BPT; ++k) { c[k] = (k < per && b0 + k < nb) ? cnt[b0 + k] : 0u; mine += c[k]; }
            const uint32_t incl = wave_incl_scan(mine);
            if (lane == 63u) wsum[wave] = incl;
            __syncthreads();
            uint32_t run = incl - mine;
            for (uint32_t k = 0; k < wave; ++k) run += wsum[k];
#pragma unroll
            for (uint32_t k = 0; k < PART_BPT; ++k)
                if (k < per && b0 + k < nb) {
                    cur[b0 + k] = run;
                    if (P64) delta64[b0 + k] = (uint64_t)G[k] - run; else delta[b0 + k] = (uint32_t)G[k] - run;
                    G[k] += c[k]; cnt[b0 + k] = 0u; run += c[k];
                }
            if (tid == PART_WG - 1u) tile_n_s = run;                     // (the last thread's running sum: the tile's records)
            __syncthreads();
            PP(2)
        }
        const uint32_t tile_n = tile_n_s;
        // ---- every record to the next slot of its bin, with its final position
#pragma unroll
        for (uint32_t j = 0; j < PART_PER / 4; ++j) {
            const uint32_t vc = tc.one ? (4u * (j * PART_WG + tid) < tc.tn ? (tc.tn - 4u * (j * PART_WG + tid) < 4u ? tc.tn - 4u * (j * PART_WG + tid) : 4u) : 0u) : (m4 >> (6u * j)) & 7u;
            const uint32_t bo = tc.one ? tc.binoff : ((m4 >> (6u * j + 3u)) & 7u) << (32u - sh);
            const uint32_t v[4] = {v4[j].x, v4[j].y, v4[j].z, v4[j].w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < vc) {
                    const uint32_t b = (v[k] >> sh) + bo;
                    const uint32_t slot = atomicAdd(&cur[b], 1u);
                    if (P64) {
                        const uint64_t p = slot + delta64[b];
                        stage[slot] = make_uint2((uint32_t)p, (v[k] & omask) | tbit | (((uint32_t)(p >> 32) << 1) << sh));
                    } else
                    stage[slot] = make_uint2(slot + delta[b], (v[k] & omask) | tbit);      // t = 1
                }
        }
        PP(3)
        __syncthreads();
        PP(4)
        // ---- the next tile is counted now (its records have landed; nobody reads the counters before the next scan) ...
        const Tile tnext = tn_;
        if (tnext.any) {
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) v4[j] = rv[j];
            m4 = mv;
            count_tile(tnext, v4, m4);
            tn_ = next_tile(tnext);
            if (tn_.any) load_tile(tn_, rv, mv);                         // ... and the one after it is on its way (in front of this tile's stores: behind them -- what helps k_part_lines -- configs[2] 468 -> 497 us: here the stores are many requests, and the loads queue behind them)
        }
        PP(5)
        // ---- ... while this one leaves LDS, four slots a lane: consecutive positions as one 16-byte store
        for (uint32_t q = tid; 4u * q < tile_n; q += PART_WG) {
            const uint4 s0 = stage4[2u * q], s1 = stage4[2u * q + 1u];   // (p0, v0, p1, v1), (p2, v2, p3, v3)
            if (P64) {                                                   // the positions' high parts ride in the records' bits above t
                const uint32_t rm = (tbit << 1) - 1u;
                const uint64_t h0 = (uint64_t)((s0.y >> sh) >> 1) << 32, h1 = (uint64_t)((s0.w >> sh) >> 1) << 32,
                               h2 = (uint64_t)((s1.y >> sh) >> 1) << 32, h3 = (uint64_t)((s1.w >> sh) >> 1) << 32;
                if (4u * q + 3u < tile_n && s1.z == s0.x + 3u && h3 == h0) {
                    u32x4u o = {s0.y & rm, s0.w & rm, s1.y & rm, s1.w & rm};
                    *reinterpret_cast<u32x4u *>(out + (h0 | s0.x)) = o;
                } else {
                    out[h0 | s0.x] = s0.y & rm;
                    if (4u * q + 1u < tile_n) out[h1 | s0.z] = s0.w & rm;
                    if (4u * q + 2u < tile_n) out[h2 | s1.x] = s1.y & rm;
                    if (4u * q + 3u < tile_n) out[h3 | s1.z] = s1.w & rm;
                }
            } else
            if (4u * q + 3u < tile_n && s1.z == s0.x + 3u) {
                u32x4u o = {s0.y, s0.w, s1.y, s1.w};
                *reinterpret_cast<u32x4u *>(out + s0.x) = o;
            } else {
                out[s0.x] = s0.y;
                if (4u * q + 1u < tile_n) out[s0.z] = s0.w;
                if (4u * q + 2u < tile_n) out[s1.x] = s1.y;
                if (4u * q + 3u < tile_n) out[s1.z] = s1.w;
            }
        }
        PP(6)
        if (!tnext.any) break;
        tc = tnext;
    }
    PP_END
}

// k_part_lines: k_part writing WHOLE 64-byte lines.  What bounds the scatter is not its instructions (the leaner kernel above runs
// no faster than round 3's) but the memory side: stores are written through, every (store instruction, 64-byte line) pair is a
// request of its own, and a request that does not cover its line is a read-modify-write at the memory: a tile's run of 7..17
// records per bin costs two of those (tools/store_bench.hip: scattered pieces below 64 bytes write at 0.4 .. 2.9 TB/s, whole
// lines at 7; WRITE_SIZE of round 3's k_part: 1.85 .. 2.06 x its records).  Here every bin keeps the records that do not fill
// a line yet -- at most 15 -- in LDS (64 bytes per bin) until the next tiles complete it: after a producer's first, aligning
// piece of a bin every store to that bin is an aligned 64-byte line (16 lanes), and each line is written once.  Per tile: count
// per bin; scan (per bin: stage cursor, records to emit = up to the last line border, lines = tasks); records to their bins'
// stage slots; one 16-lane group per line writes it from (carry, stage); the bins' owner threads move the tiles' tails into the
// carries.  Needs 64 + 18 bytes of LDS per bin next to the 32 KB stage: up to ~1500 bins (launch_part falls back to k_part).
constexpr uint32_t PL_TASKS = PART_TILE / 16u + 16u;                     // + one task per bin (a first, aligning piece)
// (tasks of a tile: its lines -- at most (PART_TILE + 15 nb) / 16 -- plus one per bin whose first piece is not aligned)
constexpr uint32_t PL_CS = 17;                                          // words per bin's carry row: 16 records on a stride that spreads the bins over the LDS banks (a stride of 16 put every bin's record i on two banks: 77 % of the LDS cycles were bank conflicts)
__host__ __device__ inline size_t part_lines_lds(uint32_t nb, bool p64 = false) { return (size_t)nb * (16u + 4u * PL_CS) + ((size_t)PL_TASKS + 2u * nb) * 4u + (p64 ? 4u * (size_t)nb : 0u); }

template <bool P64>           // P64: 64-bit positions in `out` (see k_part): the bins' cursors are 64-bit registers, a line's lanes read the high word from ghi[bin]
__global__ __launch_bounds__(PART_WG) void k_part_lines(ScanArgs a, const uint64_t *binbase, uint32_t *out)
{
    __shared__ uint4 stage4[PART_TILE / 4];                              // the tile's records, grouped by bin
    extern __shared__ __attribute__((aligned(16))) uint32_t part_lds_al[];
    uint32_t *part_lds = part_lds_al;
    __shared__ uint32_t wsum[PART_WG / 64], n_tasks_s;
    uint32_t *stage = reinterpret_cast<uint32_t *>(stage4);
    const uint32_t nb = a.n_bins, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // dg[b]: (stage start | carried records << 14 | records to emit << 18, position of the bin's next record in out) -- what a line's lanes need
    // of its bin, one 8-byte read; task[j]: bin | line of the bin << 11
    uint32_t *cnt = part_lds, *cur = cnt + nb;
    uint2 *dg = reinterpret_cast<uint2 *>(cur + nb);                     // (8-byte aligned: part_lds is, and cnt + cur are 2 nb words)
    uint32_t *cb = reinterpret_cast<uint32_t *>(dg + nb);                // [nb][PL_CS]: the bins' carried records
    uint32_t *task = cb + (size_t)nb * PL_CS;
    uint32_t *ghi = task + PL_TASKS + 2u * nb;                           // P64: high word of the bin's next position (part_lines_lds(nb, true))
    typedef typename std::conditional<P64, uint64_t, uint32_t>::type pos_t;
    const uint32_t per = (nb + PART_WG - 1u) / PART_WG, b0 = tid * per;
    constexpr uint32_t BPT = 3;                                          // bins a thread owns at most (launch_part: nb <= 3 * PART_WG)
    pos_t G[BPT]; uint32_t C[BPT];                                       // per owned bin: position of its next record in out; records carried
#pragma unroll
    for (uint32_t k = 0; k < BPT; ++k) {
        G[k] = 0u; C[k] = 0u;
        if (k < per && b0 + k < nb) { G[k] = (pos_t)binbase[b0 + k] + a.counts[(size_t)(b0 + k) * gridDim.x + blockIdx.x]; cnt[b0 + k] = 0u; }
    }
    __syncthreads();
    const uint32_t sh = a.bin_shift, omask = (1u << sh) - 1u, tbit = 1u << sh;
    const uint32_t n_seg = a.prod_waves * a.n_sub, seg0 = blockIdx.x * n_seg;
    // (the producer's segments as ONE stream cut into tiles, like k_part)
    // (one word per segment: its padded start | the segment's padding, 0 .. 3 records, in the two low bits -- a second array of counts was the 512 bytes
    // by which k_part_lines<true> at 477 bins no longer fitted a CU twice)
    __shared__ uint32_t segp_s[16u * MAX_SUB + 1u];
    for (uint32_t i = tid; i < n_seg; i += PART_WG) segp_s[i] = a.wave_cnt[seg0 + i];
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < n_seg; ++i) { const uint32_t n = segp_s[i]; segp_s[i] = run | ((0u - n) & 3u); run += (n + 3u) & ~3u; }
        segp_s[n_seg] = run;
    }
    __syncthreads();
    auto seg_p = [&](uint32_t i) { return segp_s[i] & ~3u; };
    auto seg_n = [&](uint32_t i) { const uint32_t w = segp_s[i]; return (segp_s[i + 1u] & ~3u) - (w & ~3u) - (w & 3u); };
    const uint32_t l_pad = segp_s[n_seg];                                // padded records of the producer
    // start in the stream, the segment that holds it; one: the tile's records all lie in that segment (the rule where segments are long: the tile
    // is then described by two wave-uniform words, tn records from the segment's offset v0 - start on, like rounds 3-4's tiles -- the per-group
    // meta words below cost the partition of N = 1e10 6 % when every tile used them)
    struct Tile { uint32_t v0, w0, tn, binoff; bool any, one; };
    auto tile_at = [&](uint32_t v0, uint32_t w0) {
        Tile t; t.v0 = v0; t.w0 = w0; t.any = v0 < l_pad; t.one = false; t.tn = 0u; t.binoff = 0u;
        if (t.any) {
            while (seg_p(t.w0 + 1u) <= v0) ++t.w0;
            const uint32_t end = v0 + PART_TILE < l_pad ? v0 + PART_TILE : l_pad;
            t.one = end <= seg_p(t.w0 + 1u);
            if (t.one) { const uint32_t left = seg_n(t.w0) - (v0 - seg_p(t.w0)); t.tn = left < PART_TILE ? left : PART_TILE; t.binoff = (t.w0 % a.n_sub) << (32u - sh); }
        }
        return t;
    };
    auto next_tile = [&](const Tile &c) { return tile_at(c.v0 + PART_TILE, c.w0); };
    // records 4 (j * PART_WG + tid) .. + 3 of the tile (16-byte loads); meta: per group of four how many of them are records (0 .. 4) and the
    // number of their sub-region (= high part of the cell), six bits a group
    auto load_tile = [&](const Tile &t, uint4 (&r)[PART_PER / 4], uint32_t &meta) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        meta = 0u;
        if (t.one) {                                                     // (groups past the tile's end read its last group again: never used, the passes look at tn)
            const u32x4 *src = reinterpret_cast<const u32x4 *>(a.pool + (size_t)(seg0 + t.w0) * a.cap_w + (t.v0 - seg_p(t.w0)));
            const uint32_t lastq = (t.tn - 1u) >> 2;
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) {
                const uint32_t q = j * PART_WG + tid;
                const u32x4 x = __builtin_nontemporal_load(src + (q < lastq ? q : lastq));
                r[j] = make_uint4(x.x, x.y, x.z, x.w);
            }
            return;
        }
#pragma unroll
        for (uint32_t j = 0; j < PART_PER / 4; ++j) {
            const uint32_t v = t.v0 + 4u * (j * PART_WG + tid);
            r[j] = make_uint4(0u, 0u, 0u, 0u);
            if (v < l_pad) {
                uint32_t w = t.w0;
                while (seg_p(w + 1u) <= v) ++w;
                const uint32_t off = v - seg_p(w), n = seg_n(w), vc = n - off < 4u ? n - off : 4u;
                const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.pool + (size_t)(seg0 + w) * a.cap_w + off));
                r[j] = make_uint4(x.x, x.y, x.z, x.w);
                meta |= (vc | ((w % a.n_sub) << 3)) << (6u * j);
            }
        }
    };
    auto count_tile = [&](const Tile &t, const uint4 (&r)[PART_PER / 4], uint32_t meta) {
        if (t.one) {
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) {
                const uint32_t i = 4u * (j * PART_WG + tid);
                const uint32_t v[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) if (i + k < t.tn) atomicAdd(&cnt[(v[k] >> sh) + t.binoff], 1u);
            }
            return;
        }
#pragma unroll
        for (uint32_t j = 0; j < PART_PER / 4; ++j) {
            const uint32_t vc = (meta >> (6u * j)) & 7u, bo = ((meta >> (6u * j + 3u)) & 7u) << (32u - sh);
            const uint32_t v[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) if (k < vc) atomicAdd(&cnt[(v[k] >> sh) + bo], 1u);
        }
    };
    Tile tc = tile_at(0u, 0u);
    if (!tc.any) return;
    uint4 rv[PART_PER / 4], v4[PART_PER / 4];
    uint32_t mv = 0, m4 = 0;                                             // the groups' meta words of rv / v4
    load_tile(tc, rv, mv);
#pragma unroll
    for (uint32_t j = 0; j < PART_PER / 4; ++j) v4[j] = rv[j];
    m4 = mv;
    count_tile(tc, v4, m4);
    Tile tn_ = next_tile(tc);
    if (tn_.any) load_tile(tn_, rv, mv);
    __syncthreads();                                                     // the first tile's counts are complete
    PP_DECL
    for (;;) {
        uint32_t N[BPT], S[BPT], E[BPT], Cold[BPT];
        // ---- scan: per owned bin the tile's records n, with the carried ones T; emit E = up to the last line border reached;
        // L lines = tasks.  One prefix sum over (n | L << 16).
        {
            uint32_t L[BPT], mine = 0;
            PP(0)
            // (no barrier here: the tile was counted in front of the barrier that ended the last write-out, and the carries the owner threads
            // have just moved are read by nobody before three more barriers)
            PP(1)
#pragma unroll
            for (uint32_t k = 0; k < BPT; ++k) {
                N[k] = 0u; E[k] = 0u; L[k] = 0u; Cold[k] = C[k];
                if (k < per && b0 + k < nb) {
                    N[k] = cnt[b0 + k]; cnt[b0 + k] = 0u;
                    const pos_t end = G[k] + C[k] + N[k], border = end & ~(pos_t)15u;
                    if (border > G[k]) { E[k] = (uint32_t)(border - G[k]); L[k] = (uint32_t)((border >> 4) - (G[k] >> 4)); }
                }
                mine += N[k] | (L[k] << 16);
            }
            const uint32_t incl = wave_incl_scan(mine);
            if (lane == 63u) wsum[wave] = incl;
            __syncthreads();
            uint32_t run = incl - mine;
            for (uint32_t k = 0; k < wave; ++k) run += wsum[k];
#pragma unroll
            for (uint32_t k = 0; k < BPT; ++k)
                if (k < per && b0 + k < nb) {
                    const uint32_t b = b0 + k, s0 = run & 0xFFFFu, t0_ = run >> 16;
                    S[k] = s0; cur[b] = s0; dg[b] = make_uint2(s0 | (C[k] << 14) | (E[k] << 18), (uint32_t)G[k]);
                    if (P64) ghi[b] = (uint32_t)((uint64_t)G[k] >> 32);
                    for (uint32_t i = 0; i < L[k]; ++i) task[t0_ + i] = b | (i << 11);
                    G[k] += E[k]; C[k] = C[k] + N[k] - E[k];
                    run += N[k] | (L[k] << 16);
                }
            if (tid == PART_WG - 1u) n_tasks_s = run >> 16;              // (the last thread's running sum is the total)
            __syncthreads();
            PP(2)
        }
        // ---- every record to the next stage slot of its bin
        if (tc.one) {
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) {
                const uint32_t i = 4u * (j * PART_WG + tid);
                const uint32_t v[4] = {v4[j].x, v4[j].y, v4[j].z, v4[j].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (i + k < tc.tn) stage[atomicAdd(&cur[(v[k] >> sh) + tc.binoff], 1u)] = (v[k] & omask) | tbit;      // t = 1
            }
        } else
#pragma unroll
        for (uint32_t j = 0; j < PART_PER / 4; ++j) {
            const uint32_t vc = (m4 >> (6u * j)) & 7u, bo = ((m4 >> (6u * j + 3u)) & 7u) << (32u - sh);
            const uint32_t v[4] = {v4[j].x, v4[j].y, v4[j].z, v4[j].w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < vc) stage[atomicAdd(&cur[(v[k] >> sh) + bo], 1u)] = (v[k] & omask) | tbit;      // t = 1
        }
        PP(3)
        __syncthreads();
        PP(4)
        // ---- the next tile is counted now (only the counters are touched; its records were loaded a tile ago) -- BEFORE this tile's lines are
        // stored: the wait for loaded registers is a wait for every older memory operation of the wave, and right behind the stores it was a wait
        // for their round trip (23 % of the kernel's cycles, tools/r04_part_phases.sh)
        const Tile tnext = tn_;
        if (tnext.any) {
#pragma unroll
            for (uint32_t j = 0; j < PART_PER / 4; ++j) v4[j] = rv[j];
            m4 = mv;
            count_tile(tnext, v4, m4);
        }
        PP(5)
        // ---- a line per 16-lane group: element e of the bin's stream (its carried records, then the tile's) goes to g + e
        {
            // (four lines a turn: each is a chain of dependent LDS reads -- task -> bin -> its descriptors -> the record -- and one at a
            // time the write-out was the longest phase of the kernel)
            constexpr uint32_t GRPS = PART_WG / 16u, UT = 4;
            const uint32_t grp = tid >> 4, l16 = tid & 15u, n_tasks = n_tasks_s;
            for (uint32_t j0 = grp; j0 < n_tasks; j0 += GRPS * UT) {
                uint32_t tt[UT], val[UT], gh[UT];
                pos_t pp[UT];
                uint2 dd[UT];
                bool on[UT];
#pragma unroll
                for (uint32_t u = 0; u < UT; ++u) { const uint32_t j = j0 + u * GRPS; on[u] = j < n_tasks; tt[u] = task[on[u] ? j : 0u]; }
#pragma unroll
                for (uint32_t u = 0; u < UT; ++u) { dd[u] = dg[tt[u] & 0x7FFu]; gh[u] = P64 ? ghi[tt[u] & 0x7FFu] : 0u; }
#pragma unroll
                for (uint32_t u = 0; u < UT; ++u) {
                    const uint32_t bq = tt[u] & 0x7FFu, d = dd[u].x, g = dd[u].y;
                    const uint32_t s0 = d & 0x3FFFu, c = (d >> 14) & 15u, e_n = d >> 18;
                    const uint32_t e = ((tt[u] >> 11) << 4) + l16 - (g & 15u);      // the lane's element of the bin's stream (before the first one: wraps)
                    pp[u] = P64 ? (pos_t)((((uint64_t)gh[u] << 32) | g) + e) : (pos_t)(g + e);      // (e < e_n <= 8207 where it is used: no wrap)
                    on[u] = on[u] && e < e_n;
                    const uint32_t *srcp = e < c ? cb + (bq * PL_CS + e) : stage + (s0 + e - c);
                    val[u] = on[u] ? *srcp : 0u;
                }
#pragma unroll
                for (uint32_t u = 0; u < UT; ++u) if (on[u]) __builtin_nontemporal_store(val[u], out + pp[u]);      // (whole lines, read once by the next kernel: -1.5 % against plain stores)
            }
        }

        // ---- ... and the tile after it is on its way (behind the stores: by the time its registers are waited for, both are long done)
        if (tnext.any) {
            tn_ = next_tile(tnext);
            if (tn_.any) load_tile(tn_, rv, mv);
        }
        PP(6)
        __syncthreads();                                                 // the lines have been read from the carries and the stage
        // ---- the tails into the carries: a bin that emitted keeps the last C records of the tile, one that did not appends all of them
#pragma unroll
        for (uint32_t k = 0; k < BPT; ++k)
            if (k < per && b0 + k < nb) {
                uint32_t *cbb = cb + (size_t)(b0 + k) * PL_CS;
                if (E[k]) { for (uint32_t i = 0; i < C[k]; ++i) cbb[i] = stage[S[k] + N[k] - C[k] + i]; }
                else      { for (uint32_t i = 0; i < N[k]; ++i) cbb[Cold[k] + i] = stage[S[k] + i]; }
            }
        PP(7)
        if (!tnext.any) break;
        tc = tnext;
    }
    PP_END
    // ---- the end of the producer's records: what the bins still carry (a last, partial line each)
#pragma unroll
    for (uint32_t k = 0; k < BPT; ++k)
        if (k < per && b0 + k < nb) for (uint32_t i = 0; i < C[k]; ++i) out[G[k] + i] = cb[(size_t)(b0 + k) * PL_CS + i];
}

// k_part2: second level, one workgroup per bin (bins wider than a region only): the bin's records are counted per
// 64 KB region of the table, the regions' bases go to regbase[bin * F2 + sub] (F2 = regions per bin), and a second
// sweep (served by the L2: a bin's records are a few hundred KB) moves each record to its region's range of `out`,
// tile by tile through LDS like k_part.
__global__ __launch_bounds__(PART_WG) void k_part2(const uint32_t *recs, const uint64_t *binbase, uint32_t bin_shift,
                                                   uint64_t *regbase, uint32_t *out)
{
    constexpr uint32_t F2MAX = 1u << (BIN_SHIFT_MAX - REGION_SHIFT);
    __shared__ uint2 stage[PART_TILE];
    __shared__ uint32_t cnt[F2MAX], toff[F2MAX], gcur[F2MAX];
    __shared__ uint32_t wsum[PART_WG / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t f2 = 1u << (bin_shift - REGION_SHIFT), omask = (1u << bin_shift) - 1u;
    const uint32_t bin = blockIdx.x;
    const uint64_t lo = binbase[bin], hi = binbase[bin + 1];
    for (uint32_t i = tid; i < f2; i += PART_WG) cnt[i] = 0u;
    __syncthreads();
    // ---- sweep 1: records per region.  Aligned groups of four records (16-byte loads); records outside [lo, hi)
    // read as 0 = no record (t == 0)
    {
        const uint64_t q0 = lo >> 2, q1 = (hi + 3u) >> 2;
        const uint4 *rq = reinterpret_cast<const uint4 *>(recs);
        constexpr uint32_t U = 4;
        for (uint64_t qb = q0; qb < q1; qb += (uint64_t)PART_WG * U) {
            uint4 r[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint64_t q = qb + (uint64_t)PART_WG * u + tid;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (q < q1) {
                    v = rq[q];
                    const uint64_t i = q << 2;
                    if (i < lo || i + 3u >= hi) {
                        v.x = (i >= lo && i < hi) ? v.x : 0u; v.y = (i + 1u >= lo && i + 1u < hi) ? v.y : 0u;
                        v.z = (i + 2u >= lo && i + 2u < hi) ? v.z : 0u; v.w = (i + 3u >= lo && i + 3u < hi) ? v.w : 0u;
                    }
                }
                r[u] = v;
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t wv[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) if (wv[k] >> bin_shift) atomicAdd(&cnt[(wv[k] & omask) >> REGION_SHIFT], 1u);
            }
        }
    }
    __syncthreads();
    part_scan(cnt, gcur, f2, wsum);                                  // gcur[sub] = where region sub starts inside the bin
    for (uint32_t i = tid; i < f2; i += PART_WG) { regbase[(size_t)bin * f2 + i] = lo + gcur[i]; cnt[i] = 0u; }
    __syncthreads();
    // ---- sweep 2: tile by tile into the regions' ranges (the next tile's records are loaded meanwhile)
    auto load_tile = [&](uint64_t t0, uint32_t (&v)[PART_PER]) {
        const uint32_t tn = hi - t0 < PART_TILE ? (uint32_t)(hi - t0) : PART_TILE;
#pragma unroll
        for (uint32_t j = 0; j < PART_PER; ++j) {
            const uint32_t i = j * PART_WG + tid;
            v[j] = i < tn ? recs[t0 + i] : 0u;
        }
    };
    uint32_t nxt[PART_PER];
    if (lo < hi) load_tile(lo, nxt);
    for (uint64_t t0 = lo; t0 < hi; t0 += PART_TILE) {
        const uint32_t tn = hi - t0 < PART_TILE ? (uint32_t)(hi - t0) : PART_TILE;
        uint32_t val[PART_PER], dr[PART_PER];
#pragma unroll
        for (uint32_t j = 0; j < PART_PER; ++j) {
            val[j] = nxt[j]; dr[j] = ~0u;
            if (val[j] >> bin_shift) {
                const uint32_t d = (val[j] & omask) >> REGION_SHIFT;
                dr[j] = d | (atomicAdd(&cnt[d], 1u) << 12);
            }
        }
        if (t0 + PART_TILE < hi) load_tile(t0 + PART_TILE, nxt);
        __syncthreads();
        part_scan(cnt, toff, f2, wsum);
#pragma unroll
        for (uint32_t j = 0; j < PART_PER; ++j)
            if (dr[j] != ~0u) {
                const uint32_t d = dr[j] & 0xFFFu, r = dr[j] >> 12;
                stage[toff[d] + r] = make_uint2(gcur[d] + r, val[j]);
            }
        __syncthreads();
        for (uint32_t i = tid; i < tn; i += PART_WG) { const uint2 sv = stage[i]; out[lo + sv.x] = sv.y; }
        __syncthreads();
        for (uint32_t i = tid; i < f2; i += PART_WG) { gcur[i] += cnt[i]; cnt[i] = 0u; }
        __syncthreads();
    }
}

// ---- second level without a second sweep -------------------------------------------------------------------------
// k_part2 reads a bin's records twice (count per region, then move) because every region's range of the output must be
// known before the first record moves.  k_sort_tiles does not move records between tiles at all: a bin's records are
// taken TILE by TILE (8192), each tile is sorted by region in LDS and leaves as 16-bit offsets inside the region (the
// region is what the position says), 16 KB per tile at out16[row * PART_TILE ..], row = the tile's number over all bins
// (tbase[bin] + tile of the bin); where the regions' runs start inside the tile goes to the bin's index,
// idx[(tbase[bin] + t) * (f2 + 1) + sub] (t: the tile of the bin; entry f2: the tile's record count).  k_apply_tiles
// then builds a region from its run of every tile of the bin.  One read of 4 bytes and one write of 2 per record here,
// one read of 2 there (k_part2 + k_apply: 8 + 4 and 4).
__global__ __launch_bounds__(PART_WG) void k_tile_bases(const uint64_t *binbase, uint32_t n_bins, uint32_t *tbase)
{
    __shared__ uint32_t cnt[BIN_MAX], toff[BIN_MAX];
    __shared__ uint32_t wsum[PART_WG / 64];
    for (uint32_t b = threadIdx.x; b < n_bins; b += PART_WG) cnt[b] = (uint32_t)((binbase[b + 1] - binbase[b] + PART_TILE - 1u) / PART_TILE);
    __syncthreads();
    part_scan(cnt, toff, n_bins, wsum);
    for (uint32_t b = threadIdx.x; b < n_bins; b += PART_WG) { tbase[b] = toff[b]; if (b == n_bins - 1u) tbase[n_bins] = toff[b] + cnt[b]; }
}

__global__ __launch_bounds__(PART_WG) void k_sort_tiles(const uint32_t *recs, const uint64_t *binbase, uint32_t bin_shift,
                                                        const uint32_t *tbase, uint16_t *idx, uint16_t *out16, uint32_t nt_rows)
{
    constexpr uint32_t F2MAX = 1u << (BIN_SHIFT_MAX - REGION_SHIFT);
    __shared__ uint4 stage4[PART_TILE / 8];                              // the tile's 16-bit offsets, sorted by region
    __shared__ uint32_t cnt[F2MAX], toff[F2MAX];
    __shared__ uint32_t wsum[PART_WG / 64];
    uint16_t *stage = reinterpret_cast<uint16_t *>(stage4);
    const uint32_t tid = threadIdx.x;
    const uint32_t f2 = 1u << (bin_shift - REGION_SHIFT), omask = (1u << bin_shift) - 1u;
    // every region's counter comes in R copies, a lane uses copy lane % R: 64 lanes on 32 counters is what an LDS add is slowest at (10 cycles
    // an instruction against 6.5 on 128 and more, tools/lds_bench.hip), and the ranks were a third of the kernel's cycles (tools/r04_sort_phases.sh).
    // The copies of a region lie next to each other, so the scan hands each its own piece of the region's run.
    const uint32_t rsh = f2 <= 32u ? 4u : f2 <= 64u ? 3u : f2 <= 128u ? 2u : f2 <= 256u ? 1u : 0u, nc = f2 << rsh;      // nc <= F2MAX counters
    const uint32_t mycopy = (threadIdx.x & 63u) & ((1u << rsh) - 1u);
    // workgroup (bin, k) of gridDim.y takes the bin's tiles k, k + gridDim.y, ...: tiles are independent of each other, and a
    // workgroup per BIN left the CUs unevenly loaded (477 or 1193 workgroups of 8 waves over 256 CUs, 292 on the text workload)
    const uint32_t bin = blockIdx.x, kq = blockIdx.y, nq = gridDim.y;
    const uint64_t lo = binbase[bin], hi = binbase[bin + 1];
    const uint32_t row0 = tbase[bin];
    uint16_t *bidx = idx + (size_t)row0 * (f2 + 1u);
    for (uint32_t i = tid; i < nc; i += PART_WG) cnt[i] = 0u;
    __syncthreads();
    const uint64_t step = (uint64_t)PART_TILE * nq, first = lo + (uint64_t)PART_TILE * kq;
    auto load_tile = [&](uint64_t t0, uint32_t (&v)[PART_PER]) {
        const uint32_t tn = hi - t0 < PART_TILE ? (uint32_t)(hi - t0) : PART_TILE;
#pragma unroll
        for (uint32_t j = 0; j < PART_PER; ++j) {
            const uint32_t i = j * PART_WG + tid;
            v[j] = i < tn ? __builtin_nontemporal_load(recs + t0 + i) : 0u;
        }
    };
    uint32_t nxt[PART_PER];
    if (first < hi) load_tile(first, nxt);
    uint32_t row = kq;
    __shared__ uint32_t nv_s;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    ST_DECL
    for (uint64_t t0 = first; t0 < hi; t0 += step, row += nq) {
        ST_WAITVM ST(0)
        uint32_t val[PART_PER], dr[PART_PER];
#pragma unroll
        for (uint32_t j = 0; j < PART_PER; ++j) {
            val[j] = nxt[j]; dr[j] = ~0u;
            if (val[j] >> bin_shift) {                                   // (0: no record)
                const uint32_t d = (((val[j] & omask) >> REGION_SHIFT) << rsh) | mycopy;
                dr[j] = d | (atomicAdd(&cnt[d], 1u) << 12);
            }
        }
        ST(1)
        if (t0 + step < hi) load_tile(t0 + step, nxt);
        ST(2)
        __syncthreads();
        ST(3)
        // the regions' starts inside the tile: f2 <= 512 counters, one per thread; the counters go back to zero right here (round 3 cleared them in
        // a pass of their own behind two more barriers: four barriers a tile now instead of six)
        {
            static_assert(F2MAX <= PART_WG, "a region's counter per thread");
            const uint32_t c = tid < nc ? cnt[tid] : 0u;
            const uint32_t incl = wave_incl_scan(c);
            if (lane == 63u) wsum[wave] = incl;
            __syncthreads();
            uint32_t run = incl - c;
            for (uint32_t k = 0; k < wave; ++k) run += wsum[k];
            // (the tile's f2 + 1 entries lie together -- idx[(row0 + row) * (f2 + 1) + region] --: one or a few whole lines per tile.  Until round 6 the index
            // was region-major, every tile writing f2 + 1 two-byte entries a row stride apart: 0.5 ms of k_sort_tiles' 2.85 at 512 regions per bin)
            if (tid < nc) { toff[tid] = run; cnt[tid] = 0u; if (!(tid & ((1u << rsh) - 1u))) bidx[(size_t)row * (f2 + 1u) + (tid >> rsh)] = (uint16_t)run; }
            if (tid == nc - 1u) { nv_s = run + c; bidx[(size_t)row * (f2 + 1u) + f2] = (uint16_t)(run + c); }
            __syncthreads();
        }
        ST(4)
#pragma unroll
        for (uint32_t j = 0; j < PART_PER; ++j)
            if (dr[j] != ~0u) stage[toff[dr[j] & 0xFFFu] + (dr[j] >> 12)] = (uint16_t)(val[j] & ((1u << REGION_SHIFT) - 1u));
        ST(5)
        __syncthreads();
        ST(6)
        const uint32_t nv = nv_s;                                        // records of the tile
        uint4 *dst = reinterpret_cast<uint4 *>(out16 + (size_t)(row0 + row) * ROW_STRIDE);
        // (whole 16-byte groups: the row is the tile's alone.  Non-temporal where the rows of the pass do not fit the Infinity Cache anyway -- N = 1e10
        // 1.56 -> 1.50 ms and 3 % in k_apply_tiles, configs[4]'s shape 433 -> 391 us --; where they do, plain stores leave them there for
        // k_apply_tiles: the text workload's 48 MB of rows 65 against 101 us in that kernel)
        if (nt_rows) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            for (uint32_t i = tid; i < (nv + 7u) / 8u; i += PART_WG) { const uint4 v = stage4[i]; const u32x4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(dst + i)); }
        } else
            for (uint32_t i = tid; i < (nv + 7u) / 8u; i += PART_WG) dst[i] = stage4[i];
        // (no barrier here: the next tile's ranks touch the counters only -- cleared above -- and its staging comes behind two barriers)
        ST(7)
    }
    ST_END
}

// k_apply_tiles: k_apply on the output of k_sort_tiles: the region's records are its run in every tile of its bin.
// Wave w takes the tiles w, w + 8, ... of the bin (a lane reads one tile's two index entries), then their runs one after
// the other, four 16-bit records per lane and step from 8-byte-aligned loads (and the 65th group of a run with them); the
// loads of the next four runs are in flight while four are added.
// MODE (round 5; clusterChoose without the table, ClusterBWT_DA.cpp:385-423): 0 -- the finished region is written to the table; 1 -- nothing is
// written: the region's row segments give row maxima and non-zero counts (whole rows: plain stores; rows that cross a region border: atomic max /
// add on the zeroed arrays) and the count of its last segment; 2 -- the regions are built once more and the rows that passed the host's test
// (row_off[r + 1] > row_off[r]) leave their non-zero cells as (idRef, sim) pairs in ascending idRef at pairs[row_off[r] ..] (regions without a passing
// row are skipped before a record is read).  Both need n_refs >= 256 (at most 257 row segments per 64 KB region, a wave each).
// bytes wb .. wb + 3 of a word that lie in [s, e)
__device__ __forceinline__ uint32_t keep_bytes(uint32_t x, uint32_t wb, uint32_t s, uint32_t e)
{
    uint32_t m = 0xFFFFFFFFu;
    if (wb < s) { const uint32_t d = s - wb; m = d >= 4u ? 0u : m << (8u * d); }
    if (wb + 4u > e) { const uint32_t d = e > wb ? e - wb : 0u; m &= d >= 4u ? 0xFFFFFFFFu : ((1u << (8u * d)) - 1u); }
    return x & m;
}
__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }   // bit 7 of every non-zero byte

// MODE 1 with at most FIN_SEGS row segments per region (n_refs >= 256): every thread looks at its eight 16-byte pieces of the region -- 97 % of
// them are zero on configs[2] --, finds the row segment of a non-zero piece with one multiplication by 1 / n_refs (corrected by one) and adds the
// piece's maximum / non-zero count to the segment's two LDS words; a piece that holds a row border goes byte by byte.  (The first version gave every
// segment to a wave -- 14 segments of 5000 bytes on 8 waves, each a chain of dependent LDS reads and wave reductions: 1.27 ms on configs[2]
// against 0.89 for the kernel that WRITES the table.)
constexpr uint32_t FIN_SEGS = 258;
// the segments' owner threads: whole rows are stored, rows that cross a region border added atomically (the arrays were zeroed); seg_acc is left zero
__device__ __forceinline__ void fin_region_store(uint32_t *seg_acc, const ApplyFin &f, uint32_t region, uint64_t r0, uint32_t o0, uint32_t nseg, uint32_t len)
{
    for (uint32_t j = threadIdx.x; j < nseg; j += APPLY_WG) {
        const uint32_t mx = seg_acc[j], nz = seg_acc[nseg + j];
        seg_acc[j] = 0u; seg_acc[nseg + j] = 0u;
        const uint64_t row = r0 + j, e64 = (uint64_t)(j + 1u) * f.n_refs - o0;
        const bool whole = (j != 0u || o0 == 0u) && e64 <= len;
        if (whole) { f.row_max[row] = mx; f.row_nnz[row] = nz; }
        else { if (mx) atomicMax(&f.row_max[row], mx); if (nz) atomicAdd(&f.row_nnz[row], nz); }
        if (j == nseg - 1u) f.last_nnz[region] = nz;
    }
}
__device__ __forceinline__ void fin_region_rows(uint4 *reg4, uint32_t *seg_acc, const ApplyFin &f, uint32_t region, uint64_t r0, uint32_t o0, uint32_t nseg, uint32_t len)
{
    // (seg_acc is all zero on entry: cleared at the kernel's start, and by the loop at the end of this function behind every use; the region's LDS copy
    // is left all zero too -- a piece that is looked at and not zero is zeroed right there: the next region needs no clearing pass)
    const uint32_t tid = threadIdx.x;
    const float invf = 1.0f / (float)f.n_refs;
    const uint32_t nq = (len + 15u) >> 4;
    for (uint32_t c = tid; c < nq; c += APPLY_WG) {
        const uint4 v = reg4[c];
        if (!(v.x | v.y | v.z | v.w)) continue;
        reg4[c] = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t x0 = o0 + 16u * c;                          // position of the piece's first byte counted from the start of row r0 (< 2^26)
        uint32_t seg = (uint32_t)((float)x0 * invf);
        if (seg * f.n_refs > x0) --seg; else if ((seg + 1u) * f.n_refs <= x0) ++seg;      // (float: off by one at most)
        const uint32_t border = (seg + 1u) * f.n_refs - o0;        // where the next row starts, in region bytes
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (border >= 16u * c + 16u) {                             // the whole piece lies in one row
            uint32_t m = 0, z = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t a0 = w[i] & 255u, a1 = (w[i] >> 8) & 255u, a2 = (w[i] >> 16) & 255u, a3 = w[i] >> 24;
                const uint32_t m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3, mm = m01 > m23 ? m01 : m23;
                m = mm > m ? mm : m;
                z += (uint32_t)__popc(nz_bytes(w[i]));
            }
            atomicMax(&seg_acc[seg], m); atomicAdd(&seg_acc[nseg + seg], z);
        } else {                                                   // a row border inside (n_refs >= 256: at most one)
            for (uint32_t b = 0; b < 16u; ++b) {
                const uint32_t val = (w[b >> 2] >> (8u * (b & 3u))) & 255u;
                if (!val) continue;
                const uint32_t sg = seg + (16u * c + b >= border ? 1u : 0u);
                atomicMax(&seg_acc[sg], val); atomicAdd(&seg_acc[nseg + sg], 1u);
            }
        }
    }
    __syncthreads();
    fin_region_store(seg_acc, f, region, r0, o0, nseg, len);
}

template <int MODE>
__device__ __forceinline__ void fin_region(const uint4 *reg4, const ApplyFin &f, uint32_t region, uint64_t r0, uint32_t o0, uint32_t nseg, uint32_t len)
{
    constexpr uint32_t NWV = APPLY_WG / 64;
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t reg_base = (uint64_t)region << REGION_SHIFT;
    for (uint32_t j = wave; j < nseg; j += NWV) {                 // a wave per row segment: row r0 + j, bytes [s, e) of the region
        const uint64_t row = r0 + j;
        const uint32_t s = j ? (uint32_t)((uint64_t)j * f.n_refs - o0) : 0u;
        const uint64_t e64 = (uint64_t)(j + 1u) * f.n_refs - o0;
        const uint32_t e = e64 < len ? (uint32_t)e64 : len;
        if (MODE == 1) {
            uint32_t mx = 0, nz = 0;
            for (uint32_t c = (s >> 4) + lane; 16u * c < e; c += 64u) {
                uint4 v = reg4[c];
                if (16u * c < s || 16u * c + 16u > e) {
                    v.x = keep_bytes(v.x, 16u * c, s, e); v.y = keep_bytes(v.y, 16u * c + 4u, s, e);
                    v.z = keep_bytes(v.z, 16u * c + 8u, s, e); v.w = keep_bytes(v.w, 16u * c + 12u, s, e);
                }
                if (v.x | v.y | v.z | v.w) {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        const uint32_t a0 = w[i] & 255u, a1 = (w[i] >> 8) & 255u, a2 = (w[i] >> 16) & 255u, a3 = w[i] >> 24;
                        const uint32_t m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3, m = m01 > m23 ? m01 : m23;
                        mx = m > mx ? m : mx;
                        nz += (uint32_t)__popc(nz_bytes(w[i]));
                    }
                }
            }
            mx = wave_max(mx); nz = wave_sum(nz);
            if (lane == 0) {
                const bool whole = (j != 0u || o0 == 0u) && e64 <= len;
                if (whole) { f.row_max[row] = mx; f.row_nnz[row] = nz; }
                else { if (mx) atomicMax(&f.row_max[row], mx); if (nz) atomicAdd(&f.row_nnz[row], nz); }
                if (j == nseg - 1u) f.last_nnz[region] = nz;
            }
        } else {
            const uint64_t p0 = f.row_off[row], p1 = f.row_off[row + 1u];
            if (p1 == p0) continue;                                // the read did not pass (wave-uniform)
            uint64_t run = p0;
            if (j == 0u && o0 != 0u) {                             // the row began in an earlier region: its cells there come first
                const uint32_t k0 = (uint32_t)((row * f.n_refs) >> REGION_SHIFT);
                for (uint32_t kk = k0; kk < region; ++kk) run += f.last_nnz[kk];
            }
            const uint32_t id0 = (uint32_t)(reg_base - row * f.n_refs);      // idRef of the region's byte 0 in this row (wraps for j > 0: added back below)
            for (uint32_t c0 = s >> 4; 16u * c0 < e; c0 += 64u) {
                const uint32_t c = c0 + lane;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (16u * c < e) {
                    v = reg4[c];
                    if (16u * c < s || 16u * c + 16u > e) {
                        v.x = keep_bytes(v.x, 16u * c, s, e); v.y = keep_bytes(v.y, 16u * c + 4u, s, e);
                        v.z = keep_bytes(v.z, 16u * c + 8u, s, e); v.w = keep_bytes(v.w, 16u * c + 12u, s, e);
                    }
                }
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                const uint32_t cnt = (uint32_t)(__popc(nz_bytes(w[0])) + __popc(nz_bytes(w[1])) + __popc(nz_bytes(w[2])) + __popc(nz_bytes(w[3])));
                if (!__ballot(cnt != 0u)) continue;
                const uint32_t incl = wave_incl_scan(cnt);
                uint64_t at = run + (incl - cnt);
                if (cnt) {
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i)
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t val = (w[i] >> (8u * b)) & 255u;
                            if (val) { lime_pair_t pr; pr.id_ref = id0 + 16u * c + 4u * i + b; pr.sim = val; f.pairs[at++] = pr; }
                        }
                }
                run += rl32(incl, 63);
            }
        }
    }
}

template <bool WIDE, int MODE> __global__ __launch_bounds__(APPLY_WG) void k_apply_tiles(uint8_t *sim, size_t sim_bytes, const uint16_t *recs16, const uint32_t *tbase,
                                                          const uint16_t *idx, uint32_t bin_shift, uint32_t n_regions, ApplyFin fin, uint32_t lg_in)
{
    constexpr uint32_t RW = (1u << REGION_SHIFT) / 4u;           // words per region
    constexpr uint32_t NWV = APPLY_WG / 64, UR = 4;
    __shared__ uint4 reg4[RW / 4];
    __shared__ uint32_t seg_acc[MODE == 1 ? 2 * FIN_SEGS : 2];   // MODE 1: maximum and non-zero count of the region's row segments
    // MODE 1: every wave queues the cells its adds found at 0 -- each non-zero cell of the region exactly once, as long as none wraps (a wrap raises
    // ovf_s and the region is rebuilt) -- and the look at the region is a walk over those queues instead of over 64 KB (below)
    constexpr uint32_t QW = 768;                                 // cells a wave can queue per region (more: the region is looked at piece by piece); two workgroups per CU: 64 + 12 + 2 KB each
    __shared__ uint16_t cell_q[MODE == 1 ? NWV * QW : 2];
    __shared__ uint32_t qovf_s[2];                               // by the parity of the workgroup's region count: set during a region's adds, read behind them, cleared a region later
    uint32_t par = 0;
    uint32_t qn = 0;                                             // cells in this wave's queue (wave-uniform)
    uint32_t pf = 0, po01 = 0, po23 = 0;                         // a lane's first adds of the last add4 (a bit each) and their cells, until qflush() queues them
    uint32_t *reg = reinterpret_cast<uint32_t *>(reg4);
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the runs' borders and sources stay scalar
    const uint32_t f2 = 1u << (bin_shift - REGION_SHIFT);
    // one record: + 1 modulo 256 on byte o of the region.  Fast form: ONE returning LDS add of 1 << (8 x byte) on the word -- exact as long
    // as no cell of the word passes 255 (a carry would run into its neighbour); an add that finds its cell at 255 raises the region's
    // flag, and the region is then built again with the exact form, a compare-and-swap per record (real collections never get
    // there: a cell's sum is bounded by the read length; the wrap-around fixtures and the iid generator at few reads do).
    __shared__ uint32_t ovf_s;
    bool exact = false;
    auto add_exact = [&](uint32_t o, uint32_t t = 1u) {
        const uint32_t sh = (o & 3u) * 8u;
        uint32_t *w = &reg[o >> 2];
        uint32_t seen = *w;
        for (;;) {
            const uint32_t b = ((seen >> sh) + t) & 255u;
            const uint32_t old = atomicCAS(w, seen, (seen & ~(255u << sh)) | (b << sh));
            if (old == seen) break;
            seen = old;
        }
    };
    // the four records of a lane's group [p, p + 4), those inside [fa, fe) only.  The four adds leave together and are looked at together: a
    // record outside the run adds 0 to whatever word its bits name (one add at a time behind its own branch, each waiting for its answer,
    // the adds were 77 % of the kernel's cycles at N = 1e10 and 19 .. 34 % elsewhere: tools/r04_apply_phases.sh)
    // (`on`: MODE 1 calls with all the wave's lanes and says which of them hold a group -- its queue count is wave-uniform state that a call
    // under a divergent branch would leave stale in the lanes that sat out)
    auto add4 = [&](uint32_t p, uint2 w, uint32_t fa, uint32_t fe, bool on = true) {
        const uint32_t o[4] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
        if (exact) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) if (on && p + i >= fa && p + i < fe) add_exact(o[i]);
            return;
        }
        // (the valid slots as a 4-bit mask: lo .. hi of the group; p + 4 > fa and p < fe hold for every group that gets here.  The flag is
        // also raised by an add of 0 that meets a cell at 255 -- harmless: the exact pass follows)
        const uint32_t lo = fa > p ? fa - p : 0u, hi = fe - p < 4u ? fe - p : 4u;
        const uint32_t m = on ? ((1u << hi) - 1u) & (~0u << lo) : 0u;
        uint32_t old[4], sh[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            sh[i] = (o[i] << 3) & 24u;
            old[i] = atomicAdd(&reg[o[i] >> 2], ((m >> i) & 1u) << sh[i]);
        }
        bool over = false;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) over |= __builtin_amdgcn_ubfe(old[i], sh[i], 8u) == 255u;
        if (over) ovf_s = 1u;
        if (MODE == 1) {                                          // first adds to their cells: noted here, queued by qflush() -- the callers' lanes differ, and
            pf = 0u;                                              // the queue's count is wave-uniform state that must be kept by ALL lanes
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) pf |= (uint32_t)(((m >> i) & 1u) != 0u && __builtin_amdgcn_ubfe(old[i], sh[i], 8u) == 0u) << i;
            po01 = o[0] | (o[1] << 16); po23 = o[2] | (o[3] << 16);
        }
    };
    // (called by all lanes of the wave, right behind an add4 under its condition)
    auto qflush = [&]() {
        if (MODE != 1) return;
        uint16_t *myq = cell_q + wave * QW;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const bool first = (pf >> i) & 1u;
            const uint64_t mf = __ballot(first);
            if (first) { const uint32_t at = qn + rank_in(mf); if (at < QW) myq[at] = (uint16_t)((i & 2u ? po23 : po01) >> (16u * (i & 1u))); }
            // (the count is ONE number per wave: taken through a scalar register, not a per-lane copy -- the compiler lets lanes that have no further
            // groups leave the callers' `while (__ballot(..))` loops on their own, and a lane that sat a round out would come back with a stale count and
            // write over queued cells: round 6, the run loop of grouped_runs with 16 runs an instruction lost 60 % of a dense region's cells that way)
            qn = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qn + (uint32_t)__popcll(mf)));
        }
        pf = 0u;
    };
    auto add4c = [&](bool on, uint32_t p, uint2 w, uint32_t fa, uint32_t fe) {
        if (on) add4(p, w, fa, fe);
        qflush();
    };
    static_assert(UR == 4, "a step's four runs share one pass over their groups 64 .. 79: sixteen lanes each");
    struct Step { uint2 v[UR], v2[UR], vx; uint32_t fa[UR], fe[UR], q[UR], fax, fex, qx; const uint16_t *src[UR]; };
    if (threadIdx.x == 0) { ovf_s = 0u; qovf_s[0] = 0u; qovf_s[1] = 0u; }
    // A workgroup walks regions blockIdx.x, + gridDim.x, ... (two workgroups per CU).  What a region needs before its records
    // can be read -- its bin's tile range, then its index entries -- is fetched while the region before it is worked on: a
    // workgroup per region paid that chain of dependent loads per region (configs[2]: 76 k regions of 1.6 k records each).
    const uint32_t bsh = bin_shift - REGION_SHIFT;
    auto index_of = [&](uint32_t r, uint32_t r0, uint32_t nr, uint32_t outer, uint32_t &a_, uint32_t &e_) {   // lane l: the index entries of tile outer + wave + NWV * l of the region's bin
        const uint32_t t = outer + wave + NWV * lane;
        a_ = 0u; e_ = 0u;
        if (t < nr) { const uint16_t *ia = idx + (size_t)(r0 + t) * (f2 + 1u) + (r & (f2 - 1u)); a_ = ia[0]; e_ = ia[1]; }      // (tile-major: the lanes' entries lie a tile's f2 + 1 apart; the 64 regions an XCD works on at a time share their lines)
    };
    auto runs_of = [&](uint32_t nr, uint32_t outer) {             // tiles of this round: the wave's are wave, wave + NWV, ... < left
        const uint32_t left = nr - outer;
        return left > wave ? ((left - wave + NWV - 1u) / NWV < 64u ? (left - wave + NWV - 1u) / NWV : 64u) : 0u;
    };
    auto load_step = [&](uint32_t l0, Step &s, uint32_t a_, uint32_t e_, uint32_t nl_, uint32_t row_w) {   // the first 256 records of the runs l0 .. l0 + UR of this wave (row_w: the wave's first tile row)
#pragma unroll
        for (uint32_t u = 0; u < UR; ++u) {
            const uint32_t l = l0 + u;
            s.fa[u] = l < nl_ ? rl32(a_, l) : 0u; s.fe[u] = l < nl_ ? rl32(e_, l) : 0u;
            s.q[u] = (s.fa[u] >> 2) + lane;                       // this lane's group of four records
            s.src[u] = recs16 + (size_t)(row_w + NWV * l) * ROW_STRIDE;
            s.v[u] = make_uint2(0u, 0u);
            if (s.q[u] * 4u < s.fe[u]) s.v[u] = *reinterpret_cast<const uint2 *>(s.src[u] + (size_t)s.q[u] * 4u);
            if (!WIDE) {                                          // the run's 65th group, fetched with the step
                s.v2[u] = make_uint2(0u, 0u);
                if ((s.q[u] + 64u) * 4u < s.fe[u]) s.v2[u] = *reinterpret_cast<const uint2 *>(s.src[u] + (size_t)(s.q[u] + 64u) * 4u);
            }
        }
        // WIDE (many records: chosen at the launch).  The groups 64 .. 79 of the step's four runs, sixteen lanes a run, come with the step and are added in ONE pass: a run is 256 records on
        // average at 32 regions per bin and few start on a group border, so every second run has a 65th group -- fetched when its turn came it
        // was a memory round trip, and added in a pass of its own it cost the instructions of a full pass for one or two lanes.
        if (WIDE) {
            const uint32_t ux = lane >> 4, lx = l0 + ux;
            // (the shuffles by ALL lanes, then the select: under the condition the compiler branches, and a lane reading from a lane the branch
            // has switched off gets 0 -- with 33 .. 63 runs per wave and a last step of fewer than four, the source lanes l0 + ux sit in lane groups
            // whose own run does not exist: the groups 64 .. 79 of the step's runs were dropped, silently -- a bin of 257 .. 511 tiles whose count
            // is not a multiple of 32, e.g. the N = 1e10 series' 307 tiles per bin; found in round 5 by the clustered full-size test)
            const uint32_t sa = (uint32_t)__shfl((int)a_, (int)(lx & 63u)), se = (uint32_t)__shfl((int)e_, (int)(lx & 63u));
            s.fax = lx < nl_ ? sa : 0u; s.fex = lx < nl_ ? se : 0u;
            s.qx = (s.fax >> 2) + 64u + (lane & 15u);
            s.vx = make_uint2(0u, 0u);
            if (s.qx * 4u < s.fex) s.vx = *reinterpret_cast<const uint2 *>(recs16 + (size_t)(row_w + NWV * lx) * ROW_STRIDE + (size_t)s.qx * 4u);
        }
    };
    // Short runs (round 6).  A tile row holds 8192 records of its bin, so a region's run in it has 8192 / (regions per bin) records on average: 256 at 32
    // regions per bin (the N = 1e10 series), 64 at 128 (tables of 10 GB), 16 at 512 -- and with a whole wave per run, a lane a group of four, such runs
    // keep 16 or 4 of the 64 lanes busy: the kernel's time followed the NUMBER of runs, not of records (configs[4]'s shape, clustered: 4.5 ms at 128
    // regions per bin, 7.2 at 256, 13.0 at 512 for the same 1.7e9 records).  Here 2^lg lanes share a run, a lane two groups of four: 64 >> lg runs per
    // instruction, 8 << lg records of each per pass (the mean run x 2); longer runs take further passes.
    const uint32_t lg = lg_in ? lg_in : (f2 >= 512u ? 2u : f2 == 256u ? 3u : f2 == 128u ? 4u : f2 == 64u ? 5u : 6u);      // 6: a wave per run (below)
    struct GStep { uint2 v0, v1; uint32_t fa, fe, q; const uint16_t *src; };
    auto grouped_runs = [&](uint32_t a_, uint32_t e_, uint32_t nl_, uint32_t row_w) {
        const uint32_t LG = 1u << lg, rpi = 64u >> lg, lr_in = lane >> lg, li = lane & (LG - 1u);
        auto gload = [&](uint32_t l0, GStep &s) {
            const uint32_t lr = l0 + lr_in;
            const uint32_t sa = (uint32_t)__shfl((int)a_, (int)(lr & 63u)), se = (uint32_t)__shfl((int)e_, (int)(lr & 63u));      // (by all lanes, the select behind)
            const bool valid = lr < nl_;
            s.fa = valid ? sa : 0u; s.fe = valid ? se : 0u;
            s.q = (s.fa >> 2) + 2u * li;                          // this lane's two groups of four: q, q + 1
            s.src = recs16 + (size_t)(row_w + NWV * (valid ? lr : 0u)) * ROW_STRIDE;
            s.v0 = make_uint2(0u, 0u); s.v1 = make_uint2(0u, 0u);
            if (s.q * 4u < s.fe) s.v0 = *reinterpret_cast<const uint2 *>(s.src + (size_t)s.q * 4u);
            if ((s.q + 1u) * 4u < s.fe) s.v1 = *reinterpret_cast<const uint2 *>(s.src + (size_t)(s.q + 1u) * 4u);
        };
        GStep nxt;
        gload(0u, nxt);
        for (uint32_t l0 = 0; l0 < nl_; l0 += rpi) {
            const GStep cur = nxt;
            if (l0 + rpi < nl_) gload(l0 + rpi, nxt);              // the next runs' loads go out before these are added
            add4c(cur.q * 4u < cur.fe, cur.q * 4u, cur.v0, cur.fa, cur.fe);
            add4c((cur.q + 1u) * 4u < cur.fe, (cur.q + 1u) * 4u, cur.v1, cur.fa, cur.fe);
            for (uint32_t q = cur.q + 2u * LG; __ballot(q * 4u < cur.fe); q += 2u * LG) {      // runs beyond 8 << lg records
                uint2 w0 = make_uint2(0u, 0u), w1 = make_uint2(0u, 0u);
                if (q * 4u < cur.fe) w0 = *reinterpret_cast<const uint2 *>(cur.src + (size_t)q * 4u);
                if ((q + 1u) * 4u < cur.fe) w1 = *reinterpret_cast<const uint2 *>(cur.src + (size_t)(q + 1u) * 4u);
                add4c(q * 4u < cur.fe, q * 4u, w0, cur.fa, cur.fe);
                add4c((q + 1u) * 4u < cur.fe, (q + 1u) * 4u, w1, cur.fa, cur.fe);
            }
        }
    };
    // Workgroup b runs on XCD b % 8 (round-robin dispatch), and the runs of neighbouring regions are neighbours in every tile row -- 32 to 128 bytes each
    // at 128 to 512 regions per bin: with region = b the eight XCDs each fetched the same 128-byte lines into their own L2.  Each XCD takes a block of
    // consecutive regions instead: its 64 resident workgroups work on 64 neighbouring regions at a time.
    uint32_t region = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (region >= n_regions) return;
    uint32_t row0 = tbase[region >> bsh], n_rows = tbase[(region >> bsh) + 1u] - row0;
    uint32_t a, e;
    index_of(region, row0, n_rows, 0u, a, e);
    // MODE 1, 2: the region's rows -- first row, offset of the region's first byte in it, row segments, bytes inside the table | (MODE 2: no row of
    // it passed) << 31 -- from k_region_rows' array: a scalar load per region, the next region's in flight while this one is built
    uint4 ri = make_uint4(0u, 0u, 0u, 0u);
    if (MODE != 0) ri = fin.region_rows[region];
    if (MODE == 1) for (uint32_t i = threadIdx.x; i < 2u * FIN_SEGS; i += APPLY_WG) seg_acc[i] = 0u;
    bool clean = false;                                           // the LDS copy is all zero already (MODE 1: the last region's look at it left it so)
    __syncthreads();                                              // (everybody sees the cleared flag)
    AP_DECL
    for (;;) {
        const uint32_t next = region + gridDim.x;
        const bool more = next < n_regions;
        uint32_t nrow0 = 0, nrow1 = 0;                            // the next region's tile range: needed only after this one's records
        if (more) { nrow0 = tbase[next >> bsh]; nrow1 = tbase[(next >> bsh) + 1u]; }
        uint4 nri = make_uint4(0u, 0u, 0u, 0u);
        if (MODE != 0 && more) nri = fin.region_rows[next];
        const bool skip = MODE == 2 && (ri.w >> 31) != 0u;
        if (MODE == 1 && threadIdx.x == 0) qovf_s[par ^ 1u] = 0u;  // (last read a region ago, set again only behind this region's last barrier)
        if (!skip)
        for (exact = false;; exact = true) {                      // once; twice if a cell passed 255 under the fast adds
        qn = 0u;
        if (!(MODE == 1 && clean && !exact)) {
            for (uint32_t i = threadIdx.x; i < RW / 4; i += APPLY_WG) reg4[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        AP(0)
        for (uint32_t outer = 0; outer < n_rows; outer += NWV * 64u) {
            const uint32_t nl = runs_of(n_rows, outer);
            if (outer || exact) index_of(region, row0, n_rows, outer, a, e);
            // (wave-uniform.  The modes without the table take this path for every group size, a wave per run included: with both paths compiled in they
            // pass 128 registers and a CU holds one workgroup of them instead of two)
            if (MODE != 0 || lg < 6u) { if (nl) grouped_runs(a, e, nl, row0 + outer + wave); continue; }
            Step nxt;
            if (nl) load_step(0u, nxt, a, e, nl, row0 + outer + wave);
            AP(1)
            for (uint32_t l0 = 0; l0 < nl; l0 += UR) {
                AP_WAITVM AP(2)
                Step cur = nxt;
                if (l0 + UR < nl) load_step(l0 + UR, nxt, a, e, nl, row0 + outer + wave);     // the next runs' loads go out before these are added
                AP(3)
                if (WIDE) {
#pragma unroll
                    for (uint32_t u = 0; u < UR; ++u)
                        add4c(cur.q[u] * 4u < cur.fe[u], cur.q[u] * 4u, cur.v[u], cur.fa[u], cur.fe[u]);
                    if (__ballot(cur.qx * 4u < cur.fex)) {
                        add4c(cur.qx * 4u < cur.fex, cur.qx * 4u, cur.vx, cur.fax, cur.fex);
#pragma unroll
                        for (uint32_t u = 0; u < UR; ++u) {        // runs beyond 320 records (groups from 80 on): loaded here, rare
                            const uint32_t fa = cur.fa[u], fe = cur.fe[u];
                            for (uint32_t q = cur.q[u] + 80u; __ballot(q * 4u < fe); q += 64u)
                                { uint2 wq = make_uint2(0u, 0u); if (q * 4u < fe) wq = *reinterpret_cast<const uint2 *>(cur.src[u] + (size_t)q * 4u); add4c(q * 4u < fe, q * 4u, wq, fa, fe); }
                        }
                    }
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < UR; ++u) {
                        uint32_t q = cur.q[u];
                        uint2 w = cur.v[u], w2 = cur.v2[u];
                        const uint32_t fa = cur.fa[u], fe = cur.fe[u];
                        while (__ballot(q * 4u < fe)) {
                            add4c(q * 4u < fe, q * 4u, w, fa, fe);
                            q += 64u;
                            w = w2;
                            if ((q + 64u) * 4u < fe) w2 = *reinterpret_cast<const uint2 *>(cur.src[u] + (size_t)(q + 64u) * 4u);   // (runs beyond 512 records: further groups, loaded here)
                        }
                    }
                }
                AP(4)
            }
        }
        if (MODE == 1 && qn > QW && lane == 0u) qovf_s[par] = 1u;      // (a wave found more first adds than its queue holds: this region is looked at piece by piece)
        __syncthreads();
        AP(5)
        if (exact || !ovf_s) break;
        __syncthreads();                                          // (everybody has seen the flag)
        if (threadIdx.x == 0) ovf_s = 0u;
        }
        if (MODE != 0 && !skip && fin.big_off) {                  // the long clusters' updates of this region (bucketed by region: k_bigrec_*), exact
            const uint64_t lo = fin.big_off[region], hi = fin.big_off[region + 1u];
            for (uint64_t i = lo + threadIdx.x; i < hi; i += APPLY_WG) {
                const uint64_t r = fin.bigrecs[i];
                add_exact((uint32_t)r & ((1u << REGION_SHIFT) - 1u), (uint32_t)(r >> CELL_BITS));
            }
            __syncthreads();
        }
        // the next region's index entries go out now and land while this region is written
        uint32_t na = 0, ne = 0;
        if (more) index_of(next, nrow0, nrow1 - nrow0, 0u, na, ne);
        if (MODE != 0) {
            if (!skip) {
                const uint32_t len = ri.w & 0x7FFFFFFFu;
                clean = false;
                const bool has_big = fin.big_off && fin.big_off[region + 1u] != fin.big_off[region];
                if (MODE == 1 && ri.z <= FIN_SEGS && fin.n_refs >= 16u && !exact && !has_big && qovf_s[par] == 0u) {
                    // the walk over the queued cells: final value of the cell (a byte read), the cell zeroed (a byte store: the LDS copy is left all zero),
                    // its row segment by one multiplication, two LDS adds -- about 30 instructions per 64 cells against 640 per wave for the look at all pieces
                    const uint16_t *myq = cell_q + wave * QW;
                    uint8_t *regb = reinterpret_cast<uint8_t *>(reg4);
                    const float invf = 1.0f / (float)fin.n_refs;
                    for (uint32_t i0 = 0; i0 < qn; i0 += 64u) {
                        const uint32_t i = i0 + lane;
                        if (i < qn) {
                            const uint32_t o = myq[i], val = regb[o];
                            regb[o] = 0;
                            const uint32_t x0 = ri.y + o;
                            uint32_t seg = (uint32_t)((float)x0 * invf);
                            if (seg * fin.n_refs > x0) --seg; else if ((seg + 1u) * fin.n_refs <= x0) ++seg;
                            if (val) { atomicMax(&seg_acc[seg], val); atomicAdd(&seg_acc[ri.z + seg], 1u); }
                        }
                    }
                    __syncthreads();
                    fin_region_store(seg_acc, fin, region, (uint64_t)ri.x, ri.y, ri.z, len);
                    clean = true;
                }
                else if (MODE == 1 && ri.z <= FIN_SEGS && fin.n_refs >= 16u) {
                    fin_region_rows(reg4, seg_acc, fin, region, (uint64_t)ri.x, ri.y, ri.z, len); clean = true;
                }
                else fin_region<MODE>(reg4, fin, region, (uint64_t)ri.x, ri.y, ri.z, len);
            }
        } else {
        const size_t reg_base = (size_t)region << REGION_SHIFT;  // regions start inside the table
        uint4 *dst = reinterpret_cast<uint4 *>(sim + reg_base);
        const size_t left16 = (sim_bytes - reg_base) / 16u;      // sim_bytes is a multiple of 16
        constexpr uint32_t NST = RW / 4 / APPLY_WG;
        static_assert(RW / 4 % APPLY_WG == 0, "whole rounds of 16-byte stores");
        if (left16 >= RW / 4) {
            // the thread's eight 16-byte pieces: read together, then stored together (one after the other every piece waited for its LDS
            // read; named registers, not an array: the array went to scratch memory)
            static_assert(NST == 8, "eight 16-byte stores per thread below");
            uint4 *rp = reg4 + threadIdx.x, *dp = dst + threadIdx.x;
#define LIME_RD(J) const uint4 o##J = rp[J * APPLY_WG];
            LIME_RD(0) LIME_RD(1) LIME_RD(2) LIME_RD(3) LIME_RD(4) LIME_RD(5) LIME_RD(6) LIME_RD(7)
#undef LIME_RD
            // (non-temporal: the table is written once and not read again by the pass -- configs[2] 0.98 -> 0.91 ms, the text workload 83 -> 64 us,
            // configs[4]'s shape 2.06 -> 1.93 ms against plain stores, ABAB in one run)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LIME_ST(J) { const u32x4 x = {o##J.x, o##J.y, o##J.z, o##J.w}; __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(dp + J * APPLY_WG)); }
            LIME_ST(0) LIME_ST(1) LIME_ST(2) LIME_ST(3) LIME_ST(4) LIME_ST(5) LIME_ST(6) LIME_ST(7)
#undef LIME_ST
        } else {                                                  // the table's last region, cut short (it is its workgroup's last one)
            for (uint32_t i = threadIdx.x; i < left16; i += APPLY_WG) dst[i] = reg4[i];
        }
        }
        AP(6)
        if (!more) break;
        __syncthreads();                                          // (the region's LDS copy has been read: it may be cleared)
        ri = nri; par ^= 1u;
        AP(7)
        region = next; row0 = nrow0; n_rows = nrow1 - nrow0; a = na; e = ne;
    }
    AP_END
}

// k_apply: one workgroup builds one 64 KB region of the table in LDS -- zero, add the region's records (exact
// modulo 256 per byte cell: an LDS compare-and-swap on the containing word), write it out once with 16-byte
// stores.  Two workgroups fit a CU, so one region's write-out overlaps the next one's accumulation.  The table
// needs no clearing beforehand: every byte of it is written here.  Record: offset in its bin | t << bin_shift.
__global__ __launch_bounds__(APPLY_WG) void k_apply(uint8_t *sim, size_t sim_bytes, const uint32_t *recs, const uint64_t *regbase,
                                                    uint32_t bin_shift)
{
    constexpr uint32_t RW = (1u << REGION_SHIFT) / 4u;           // words per region
    __shared__ uint4 reg4[RW / 4];
    uint32_t *reg = reinterpret_cast<uint32_t *>(reg4);
    const uint32_t region = blockIdx.x;
    const size_t reg_base = (size_t)region << REGION_SHIFT;      // grid = regions that start inside the table
    for (uint32_t i = threadIdx.x; i < RW / 4; i += APPLY_WG) reg4[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const uint64_t lo = regbase[region], hi = regbase[region + 1];
    const uint32_t rmask = (1u << REGION_SHIFT) - 1u;
    constexpr uint32_t U = 4;
    for (uint64_t i0 = lo; i0 < hi; i0 += (uint64_t)APPLY_WG * U) {
        uint32_t r[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)APPLY_WG * u + threadIdx.x;
            r[u] = i < hi ? recs[i] : 0u;                        // t == 0: no record
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t t = r[u] >> bin_shift;
            if (t != 0u) {
                const uint32_t o = r[u] & rmask, sh = (o & 3u) * 8u;
                uint32_t *w = &reg[o >> 2];
                uint32_t seen = *w;
                for (;;) {
                    const uint32_t b = ((seen >> sh) + t) & 255u;
                    const uint32_t old = atomicCAS(w, seen, (seen & ~(255u << sh)) | (b << sh));
                    if (old == seen) break;
                    seen = old;
                }
            }
        }
    }
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(sim + reg_base);
    const size_t left = (sim_bytes - reg_base) / 16u;            // sim_bytes is a multiple of 16
    for (uint32_t i = threadIdx.x; i < RW / 4 && i < left; i += APPLY_WG) dst[i] = reg4[i];     // (non-temporal stores here: no gain, tools/r03_ab2.sh)
}

// =========================================================================================
// k_score_list: scores clusters given as (pStart,len) records.  Each wave gathers 64 clusters
// (each <= SMALL_MAX long) side by side into its LDS, files them under their length class and
// runs the same scoring routines as the scan; longer clusters are pushed to the big list.
// =========================================================================================
// BIN (round 4; device-resident arrays, the table to be built from scratch): the updates leave as 4-byte records through the
// same drains, line buffers and per-bin counts as the scan's (a workgroup = one producer of k_part), instead of
// compare-and-swaps at the memory-side atomic rate -- the two-program flow on text-like density: 4.3 ms per 1e8 symbols.
template <int EBWT, int BIN>
__global__ __launch_bounds__(SCAN_WG) void k_score_list(ScanArgs a, const lime_cluster_t *list, uint64_t n_list)
{
    __shared__ WaveLds<64 * SMALL_MAX> lds[SCAN_WG / 64];
    __shared__ uint32_t c_off[SCAN_WG / 64][65];
    __shared__ uint32_t hist[BIN ? BIN_MAX : 1], bq_sub_n[SCAN_WG / 64][MAX_SUB + 1], bq_lfill[SCAN_WG / 64][2], bq_lbuf[SCAN_WG / 64][BIN ? 2 * LBUF : 2], wg_done_s;
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: LDS bases stay scalar
    WaveLds<64 * SMALL_MAX> &L = lds[wave];
    uint32_t *off = c_off[wave];
    const uint64_t n_batches = (n_list + 63u) / 64u, stride = (uint64_t)gridDim.x * (SCAN_WG / 64);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t acc_upd = 0;
    UpdQueue qu; qu.qr = L.q_read; qu.qg = L.q_gen; qu.n = 0; qu.cap = QCAP;
    if (BIN) {
        for (uint32_t i = threadIdx.x; i < BIN_MAX; i += SCAN_WG) hist[i] = 0u;
        if (threadIdx.x == 0) wg_done_s = 0u;
        const uint32_t wave_gid = blockIdx.x * (SCAN_WG / 64) + wave;
        qu.async = true; qu.binned = true;
        qu.out = cold(a).pool + (size_t)wave_gid * cold(a).n_sub * cold(a).cap_w; qu.hist = hist;
        qu.sub_n = (lds_vu32 *)bq_sub_n[wave]; qu.lbuf = bq_lbuf[wave]; qu.lfill = (lds_vu32 *)bq_lfill[wave];
        if (lane <= MAX_SUB) bq_sub_n[wave][lane] = 0u;               // ([MAX_SUB]: the producer group's offset in the histogram: one group)
        if (lane < 2u) bq_lfill[wave][lane] = 0u;
        __syncthreads();
    }
    for (uint64_t b = (uint64_t)blockIdx.x * (SCAN_WG / 64) + wave; b < n_batches; b += stride) {
        const uint64_t c = b * 64u + lane;
        uint64_t ps = 0, len = 0;
        if (c < n_list) { ps = list[c].pStart - a.pos_base; len = list[c].len; }      // pos_base: where the arrays handed over start in the collection
        const bool bad = (len > LIME_MAX_CLUSTER) || (ps > a.n_avail) || (len > a.n_avail - ps);
        if (bad) { atomicOr(&a.stats->flags, len > LIME_MAX_CLUSTER ? LIME_FLAG_MAXLEN : LIME_FLAG_BADCLUSTER); len = 0; }
        if (len > SMALL_MAX) {
            const uint32_t k = atomicAdd(&a.stats->n_big, 1u);
            if (k < a.big_cap) { a.big[k].pStart = ps; a.big[k].len = len; }
            len = 0;
        }
        if (len < 2u) len = 0;                     // a 0/1-symbol cluster cannot hold a read and a genome
        const uint32_t len32 = (uint32_t)len;
        uint32_t incl = len32;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if ((int)lane >= d) incl += y; }
        const uint32_t my_off = incl - len32, total = rl32(incl, 63);
        off[lane] = my_off;
        if (lane == 63u) off[64] = total;
        for (uint32_t i0 = 0; i0 < total; i0 += 64u) {
            const uint32_t i = i0 + lane < total ? i0 + lane : total - 1u;
            uint32_t lo = 0, hi = 63u;                // last cluster with off <= i
            while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (off[mid] <= i) lo = mid; else hi = mid - 1u; }
            const uint64_t g = shfl64(ps, (int)lo) + (i - off[lo]);
            const uint32_t d = a.da[g];
            uint32_t f = EBWT ? sym_index(a.ebwt[g]) : 0u;
            f |= (d < a.n_reads) ? F_READ : F_GEN;
            L.da[i] = d; L.fl[i] = (uint8_t)f;
            const uint64_t rm = __ballot(d < a.n_reads && i0 + lane < total);
            if (lane == 0) *reinterpret_cast<u64a *>(&L.rb[i0 >> 3]) = rm;
        }
        const bool cA = len32 >= 2u;
        const uint16_t item = (uint16_t)(my_off | ((len32 - 1u) << 12));
        const uint64_t mA = __ballot(cA);
        if (cA) L.listA[(uint32_t)__popcll(mA & lt)] = item;
        acc_upd += score_lists<EBWT>(L, qu, a, (uint32_t)__popcll(mA));
    }
    if (BIN) {                                                        // the records still waiting for their line; this wave's counts; the workgroup's histogram
        const uint32_t n_sub = cold(a).n_sub, cap_w = cold(a).cap_w;
        drain_bin(qu, a, true);
        if (lane < n_sub) {
            const uint32_t n = qu.sub_n[lane];
            cold(a).wave_cnt[(size_t)(blockIdx.x * (SCAN_WG / 64) + wave) * n_sub + lane] = n < cap_w ? n : cap_w;
            atomicMax(&cold(a).stats->wave_records_max, n);
            if (n > cap_w) atomicOr(&cold(a).stats->flags, LIME_FLAG_POOL_FULL);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(&wg_done_s, 1u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        if (old == SCAN_WG / 64 - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (uint32_t bb = lane; bb < cold(a).n_bins; bb += 64u) cold(a).counts[(size_t)bb * gridDim.x + blockIdx.x] = hist[bb];
        }
    }
    const uint32_t tu = wave_sum(acc_upd);
    if (lane == 0 && tu) atomicAdd(&a.stats->n_updates, (unsigned long long)tu);
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
                    if (g >= a.n_refs) atomicOr(&a.stats->flags, LIME_FLAG_DOCID);
                    else if (a.sim) { sim_add(a.sim, row + g, t); ++nupd; }
                    else {                                     // records for the owner of the cell (owner-partitioned exchange)
                        const uint32_t k = atomicAdd(a.bigrec_n, 1u);
                        if (k < a.bigrec_cap) a.bigrec[k] = (row + g) | ((uint64_t)t << CELL_BITS);
                        else atomicOr(&a.stats->flags, LIME_FLAG_OVERFLOW);
                        ++nupd;
                    }
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
    // max and non-zero count of the four bytes of a word, SWAR (no per-byte bounds checks: the caller
    // has cleared the bytes outside the row)
    auto word = [](uint32_t v, uint32_t &mx, uint32_t &nz) {
        const uint32_t b0 = v & 255u, b1 = (v >> 8) & 255u, b2 = (v >> 16) & 255u, b3 = v >> 24;
        const uint32_t m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3, m = m01 > m23 ? m01 : m23;
        mx = m > mx ? m : mx;
        uint32_t t = v | (v >> 4); t |= t >> 2; t |= t >> 1;
        nz += (uint32_t)__popc(t & 0x01010101u);
    };
    for (uint64_t r = (uint64_t)blockIdx.x * (WGSZ / 64) + (threadIdx.x >> 6); r < n_reads; r += waves) {
        const uint64_t b0 = r * n_refs, b1 = b0 + n_refs;
        const uint64_t q0 = b0 >> 4, q1 = (b1 + 15ull) >> 4;            // 16-byte groups covering the row
        uint32_t mx = 0, nz = 0;
        for (uint64_t q = q0 + lane; q < q1; q += 64u) {
            uint4 v = reinterpret_cast<const uint4 *>(sim)[q];            // the table is padded to 16 bytes
            if (q == q0 || q + 1 == q1) {                                 // first / last group: clear foreign bytes
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint64_t byte = q * 16ull + k;
                    if (byte < b0 || byte >= b1) w[k >> 2] &= ~(255u << (8 * (k & 3)));
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            word(v.x, mx, nz); word(v.y, mx, nz); word(v.z, mx, nz); word(v.w, mx, nz);
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
// k_gather_pairs: the non-zero cells of the selected table rows, compacted in ascending idRef
// (the (idRef, sim) list of clusterChoose, ClusterBWT_DA.cpp:390-402).  One wave per row; cells of
// row r go to pairs[row_off[r] .. row_off[r+1]) (an empty range skips the row).
// =========================================================================================
__global__ __launch_bounds__(WGSZ) void k_gather_pairs(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs,
                                                       const uint64_t *row_off, lime_pair_t *pairs)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t waves = (uint64_t)gridDim.x * (WGSZ / 64);
    for (uint64_t r = (uint64_t)blockIdx.x * (WGSZ / 64) + (threadIdx.x >> 6); r < n_reads; r += waves) {
        uint64_t out = row_off[r];
        if (row_off[r + 1] == out) continue;
        const uint64_t b0 = r * n_refs, b1 = b0 + n_refs;
        const uint64_t w0 = b0 >> 2, w1 = (b1 + 3ull) >> 2;
        for (uint64_t wb = w0; wb < w1; wb += 64u) {
            const uint64_t w = wb + lane;
            uint32_t v = w < w1 ? reinterpret_cast<const uint32_t *>(sim)[w] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                      // bytes outside the row do not count
                const uint64_t byte = w * 4ull + k;
                if (byte < b0 || byte >= b1) v &= ~(255u << (8 * k));
            }
            const uint32_t nz = (uint32_t)((v & 0xFFu) != 0u) + (uint32_t)((v & 0xFF00u) != 0u) +
                                (uint32_t)((v & 0xFF0000u) != 0u) + (uint32_t)((v >> 24) != 0u);
            const uint32_t incl = wave_incl_scan(nz);
            uint64_t o = out + incl - nz;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t x = (v >> (8 * k)) & 255u;
                if (x) { lime_pair_t pr; pr.id_ref = (uint32_t)(w * 4ull + k - b0); pr.sim = x; pairs[o++] = pr; }
            }
            out += rl32(incl, 63);
        }
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

// blockIdx.y: one of gridDim.y arrays of n words, `pitch` words apart (k_score_big's scratch tables: one launch for all of them)
__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v, size_t pitch)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    p += (size_t)blockIdx.y * pitch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// =========================================================================================
// Owner-partitioned exchange of table updates (several GPUs, large tables): every rank leaves its updates as records
// grouped by table bin (k_part); the owner of a range of bins receives, from every rank, the slice of records of its
// bins and builds ITS block of the table alone.  k_regroup: the received slices (source-major, each grouped by bin) into
// one array grouped by bin -- a workgroup per bin copies the sources' runs one after the other.
// srcoff[s * (nb + 1) + b]: where source s's records of local bin b start in rx; dstbase[b]: where bin b starts in dst.
// =========================================================================================
__global__ __launch_bounds__(256) void k_regroup(const uint32_t *rx, const uint64_t *srcoff, uint32_t n_src, uint32_t nb,
                                                 const uint64_t *dstbase, uint32_t *dst)
{
    const uint32_t b = blockIdx.x;
    uint64_t at = dstbase[b];
    for (uint32_t s = 0; s < n_src; ++s) {
        const uint64_t lo = srcoff[(size_t)s * (nb + 1u) + b], hi = srcoff[(size_t)s * (nb + 1u) + b + 1u];
        for (uint64_t i = lo + threadIdx.x; i < hi; i += 256u) dst[at + (i - lo)] = rx[i];
        at += hi - lo;
    }
}

// the long clusters' updates of ALL ranks (cell | t << CELL_BITS): the ones that fall into this rank's block are added
// to it (exact modulo 256 per byte cell, like k_score_big on a whole table)
__global__ __launch_bounds__(256) void k_apply_bigrecs(const uint64_t *recs, uint64_t n, uint64_t cell_lo, uint64_t cell_hi, uint8_t *block)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
        const uint64_t r = recs[i], cell = r & ((1ull << CELL_BITS) - 1ull);
        if (cell >= cell_lo && cell < cell_hi) sim_add(block, cell - cell_lo, (uint32_t)(r >> CELL_BITS));
    }
}

// ---- launch wrappers (host) ------------------------------------------------------------
template <typename K> static uint32_t resident_blocks(K kernel, int block)
{
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1024u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    return (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount;
}

// ID: one per instantiation of k_scan (they share one function type, so K alone would share the static)
// per-device caches of the launch wrappers (one process may drive several GPUs from several host threads)
constexpr int MAX_DEV = 64;
static int cur_device() { int d = 0; (void)hipGetDevice(&d); return d >= 0 && d < MAX_DEV ? d : 0; }

// (probe_shift: only every 2^probe_shift-th chunk is scanned -- the density probe)
template <int ID, int WG, typename K> static uint32_t scan_grid_of(K kernel, uint32_t n_tiles, uint32_t max_blocks, uint32_t probe_shift = 0)
{
    constexpr int SCANK_WG = WG;
    static std::atomic<uint32_t> resident_of[MAX_DEV];
    std::atomic<uint32_t> &slot = resident_of[cur_device()];
    uint32_t resident = slot.load(std::memory_order_relaxed);
    if (!resident) { resident = resident_blocks(kernel, SCANK_WG); slot.store(resident, std::memory_order_relaxed); }
    const uint32_t chunks = (n_tiles + SCANK_WG / 64 - 1) / (SCANK_WG / 64), want = (chunks + (1u << probe_shift) - 1u) >> probe_shift, cap = max_blocks ? max_blocks : resident;
    const uint32_t grid = want < cap ? want : cap;
    return grid ? grid : 1u;
}

template <int ID, int WG, typename K> static void launch_scan_kernel(K kernel, const ScanArgs &a, uint32_t max_blocks, hipStream_t st)
{
    // persistent grid: as many workgroups as fit the device at once (per instantiation), or fewer for short inputs
    const uint32_t grid = scan_grid_of<ID, WG>(kernel, a.n_tiles, max_blocks, a.probe_shift);
    const uint32_t pct = a.static_pct > 100u ? 100u : a.static_pct;
    ScanArgs b = a;
    const uint64_t chunks = (((uint64_t)a.n_tiles + WG / 64 - 1) / (WG / 64) + (1u << a.probe_shift) - 1u) >> a.probe_shift;
    const uint64_t rounds = chunks / grid;             // whole rounds of chunks
    b.n_static = (uint32_t)(rounds * (uint64_t)pct / 100u);
    if (!b.n_static) b.n_static = 1u;                  // round 0 is always the workgroups' own chunks (grid <= number of chunks)
    if (b.n_static > 0xFFFFFFu) b.n_static = 0xFFFFFFu;
    b.n_static |= a.probe_shift << 24;                 // (the kernel's take() splits the word: one SGPR for both)
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WG), 0, st, b);
}

static uint32_t apply_tiles_grid(uint32_t n_regions);
// lime_init: every kernel's code is loaded and the launch wrappers' per-device figures (resident workgroups, LDS attributes) are worked out
// now, not inside a context's first pass (LiME_paired.sh runs a collection once: the first pass IS the run; on configs[1] the first launches'
// lazy loading and occupancy queries were 0.2 ms of a 0.25 ms pass)
void launch_preload()
{
    static std::atomic<bool> done[MAX_DEV];
    std::atomic<bool> &d = done[cur_device()];
    if (d.load(std::memory_order_relaxed)) return;
    for (int e = 0; e < 2; ++e) for (int b = 0; b < 3; ++b) (void)scan_grid(e, 0, b, 1u << 20, 0, 0);
    (void)scan_grid(0, 1, 0, 1u << 20, 0, 0);
    (void)apply_tiles_grid(1u << 20);
    hipFuncAttributes fa;
#define LIME_PRELOAD(K) (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(K));
    LIME_PRELOAD(k_resolve_open) LIME_PRELOAD(k_rowscan_resolve) LIME_PRELOAD(k_bin_bases) LIME_PRELOAD(k_resolve<1>) LIME_PRELOAD(k_emit) LIME_PRELOAD(k_scan_tiles) LIME_PRELOAD(k_bin_rowscan)
    LIME_PRELOAD((k_part<PART_WG, BIN_MAX, false>)) LIME_PRELOAD((k_part<PART_WG, BIN_MAX, true>)) LIME_PRELOAD(k_part_lines<false>) LIME_PRELOAD(k_part_lines<true>)
    LIME_PRELOAD(k_tile_bases) LIME_PRELOAD(k_sort_tiles) LIME_PRELOAD((k_apply_tiles<true, 0>)) LIME_PRELOAD(k_apply) LIME_PRELOAD(k_score_big<0>) LIME_PRELOAD(k_score_big<1>)
    LIME_PRELOAD(k_choose) LIME_PRELOAD(k_gather_pairs) LIME_PRELOAD((k_score_list<0, 0>)) LIME_PRELOAD((k_score_list<1, 0>))
#undef LIME_PRELOAD
    (void)hipGetLastError();
    d.store(true, std::memory_order_relaxed);
}

uint32_t scan_grid(int ebwt, int mode, int binned, uint32_t n_tiles, uint32_t max_blocks, uint32_t probe_shift)
{
    if (mode != 0) return scan_grid_of<2, ScanCfg<0, 0>::wg>(k_scan<0, 1, 0>, n_tiles, max_blocks);
    if (binned == 2) return ebwt ? scan_grid_of<6, ScanCfg<1, 2>::wg>(k_scan<1, 0, 2>, n_tiles, max_blocks, probe_shift) : scan_grid_of<5, ScanCfg<0, 2>::wg>(k_scan<0, 0, 2>, n_tiles, max_blocks, probe_shift);
    if (binned) return ebwt ? scan_grid_of<4, ScanCfg<1, 1>::wg>(k_scan<1, 0, 1>, n_tiles, max_blocks, probe_shift) : scan_grid_of<3, ScanCfg<0, 1>::wg>(k_scan<0, 0, 1>, n_tiles, max_blocks, probe_shift);
    return ebwt ? scan_grid_of<1, ScanCfg<1, 0>::wg>(k_scan<1, 0, 0>, n_tiles, max_blocks) : scan_grid_of<0, ScanCfg<0, 0>::wg>(k_scan<0, 0, 0>, n_tiles, max_blocks);
}

void launch_rowscan_resolve(const ScanArgs &a, uint32_t *counts, uint32_t *totals, uint32_t n_bins, uint32_t n_prod, hipStream_t st)
{
    hipLaunchKernelGGL(k_rowscan_resolve, dim3((n_bins + 3u) / 4u + RESOLVE_BLOCKS), dim3(256), 0, st, a, counts, totals, n_bins, n_prod);
}
void launch_bin_bases(const uint32_t *totals, uint64_t *binbase, uint32_t *tbase, uint32_t n_bins, hipStream_t st)
{
    hipLaunchKernelGGL(k_bin_bases, dim3(1), dim3(1024), 0, st, totals, binbase, tbase, n_bins, PART_TILE);
}

void launch_bin_rowscan(uint32_t *counts, uint32_t *totals, uint32_t n_bins, uint32_t n_prod, hipStream_t st)
{
    hipLaunchKernelGGL(k_bin_rowscan, dim3((n_bins + 3u) / 4u), dim3(256), 0, st, counts, totals, n_bins, n_prod);
}

void launch_part(const ScanArgs &a, uint32_t n_prod, const uint64_t *binbase, uint32_t *out, hipStream_t st, bool p64, bool lines_ok)
{
    static std::atomic<bool> attr_set[MAX_DEV];              // the attribute is per device
    std::atomic<bool> &set = attr_set[cur_device()];
    if (!set.load(std::memory_order_relaxed)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_part<PART_WG, BIN_MAX, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BIN_MAX * 12u));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_part<PART_WG, BIN_MAX, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BIN_MAX * 16u));
        set.store(true, std::memory_order_relaxed);
    }
    // whole-line writes (k_part_lines) wherever the bins' line buffers fit the LDS next to the stage; LIME_PART_LINES=0: comparison runs
    static std::atomic<uint32_t> lines_room[MAX_DEV][2];     // dynamic LDS k_part_lines may ask for on this device (0: not asked yet)
    const size_t lds_lines = part_lines_lds(a.n_bins, p64);
    const void *kl = p64 ? reinterpret_cast<const void *>(k_part_lines<true>) : reinterpret_cast<const void *>(k_part_lines<false>);
    if (lines_ok && a.n_bins <= 3u * PART_WG) {
        std::atomic<uint32_t> &room = lines_room[cur_device()][p64 ? 1 : 0];
        uint32_t r = room.load(std::memory_order_relaxed);
        if (!r) {
            hipFuncAttributes fa;
            r = 1u;
            if (hipFuncGetAttributes(&fa, kl) == hipSuccess && fa.sharedSizeBytes < 160u * 1024u) {
                const uint32_t dyn = 160u * 1024u - (uint32_t)fa.sharedSizeBytes;
                if (hipFuncSetAttribute(kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) == hipSuccess) r = dyn;
            }
            (void)hipGetLastError();
            room.store(r, std::memory_order_relaxed);
        }
        // two workgroups per CU must fit (the kernel is a chain of short phases: alone on a CU it is slower than k_part -- configs[2], 1193 bins:
        // 0.72 against 0.48 ms; N = 1e10, 477 bins, two per CU: 3.8 against 4.2 .. 4.7 ms)
        const size_t stat = 160u * 1024u - (r > 1u ? r : 0u);               // the kernel's static LDS
        if (lds_lines <= r && 2u * (lds_lines + stat + 512u) <= 160u * 1024u) {
            if (p64) hipLaunchKernelGGL(k_part_lines<true>, dim3(n_prod), dim3(PART_WG), lds_lines, st, a, binbase, out);
            else     hipLaunchKernelGGL(k_part_lines<false>, dim3(n_prod), dim3(PART_WG), lds_lines, st, a, binbase, out);
            return;
        }
    }
    if (p64) hipLaunchKernelGGL((k_part<PART_WG, BIN_MAX, true>), dim3(n_prod), dim3(PART_WG), (size_t)a.n_bins * 16u, st, a, binbase, out);
    else     hipLaunchKernelGGL((k_part<PART_WG, BIN_MAX, false>), dim3(n_prod), dim3(PART_WG), (size_t)a.n_bins * 12u, st, a, binbase, out);
}

void launch_part2(const uint32_t *recs, const uint64_t *binbase, uint32_t n_bins, uint32_t bin_shift, uint64_t *regbase,
                  uint32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_part2, dim3(n_bins), dim3(PART_WG), 0, st, recs, binbase, bin_shift, regbase, out);
}

void launch_apply(uint8_t *sim, size_t sim_bytes, const uint32_t *recs, const uint64_t *regbase, uint32_t bin_shift, hipStream_t st)
{
    const uint32_t grid = (uint32_t)((sim_bytes + ((size_t)1 << REGION_SHIFT) - 1) >> REGION_SHIFT);
    hipLaunchKernelGGL(k_apply, dim3(grid), dim3(APPLY_WG), 0, st, sim, sim_bytes, recs, regbase, bin_shift);
}

// second level by tiles (k_sort_tiles + k_apply_tiles).  tbase: n_bins + 1 words; idx: (tiles + n_bins) * (f2 + 1) 16-bit entries;
// out16: PART_TILE 16-bit records per tile row (tiles_bound() rows at most)
void launch_sort_tiles(const uint32_t *recs, const uint64_t *binbase, uint32_t n_bins, uint32_t bin_shift, uint32_t *tbase, uint16_t *idx, uint16_t *out16,
                       hipStream_t st, bool big_rows, bool tbase_ready)
{
    if (!tbase_ready) hipLaunchKernelGGL(k_tile_bases, dim3(1), dim3(PART_WG), 0, st, binbase, n_bins, tbase);
    // enough workgroups to fill the device evenly: about 8 per CU (two are resident at a time)
    const uint32_t per_bin = n_bins >= 2048u ? 1u : (2048u + n_bins - 1u) / n_bins;
    hipLaunchKernelGGL(k_sort_tiles, dim3(n_bins, per_bin), dim3(PART_WG), 0, st, recs, binbase, bin_shift, tbase, idx, out16, big_rows ? 1u : 0u);
}

// option apply_group (comparison runs): lanes per run of k_apply_tiles as a power of two, 1 .. 6 (6: a wave per run); 0: by the regions per bin
static std::atomic<uint32_t> g_apply_group{0};
void set_apply_group(uint32_t lg) { g_apply_group.store(lg >= 1u && lg <= 6u ? lg : 0u, std::memory_order_relaxed); }
static uint32_t apply_tiles_grid(uint32_t n_regions)
{
    static std::atomic<uint32_t> resident_of[MAX_DEV];           // workgroups that fit the device at once (two per CU: 64 KB of LDS each)
    std::atomic<uint32_t> &slot = resident_of[cur_device()];
    uint32_t resident = slot.load(std::memory_order_relaxed);
    if (!resident) { resident = resident_blocks(k_apply_tiles<false, 0>, APPLY_WG); slot.store(resident, std::memory_order_relaxed); }
    const uint32_t grid = n_regions < resident ? n_regions : resident;
    return grid ? grid : 1u;
}

void launch_apply_tiles_fin(int mode, size_t sim_bytes, uint32_t bin_shift, const uint32_t *tbase, const uint16_t *idx, const uint16_t *out16, bool many_records,
                            const ApplyFin &fin, hipStream_t st)
{
    const uint32_t n_regions = (uint32_t)((sim_bytes + ((size_t)1 << REGION_SHIFT) - 1) >> REGION_SHIFT);
    const dim3 grid(apply_tiles_grid(n_regions)), wg(APPLY_WG);
    if (mode == 1) {
        if (many_records) hipLaunchKernelGGL((k_apply_tiles<true, 1>), grid, wg, 0, st, nullptr, sim_bytes, out16, tbase, idx, bin_shift, n_regions, fin, g_apply_group.load(std::memory_order_relaxed));
        else              hipLaunchKernelGGL((k_apply_tiles<false, 1>), grid, wg, 0, st, nullptr, sim_bytes, out16, tbase, idx, bin_shift, n_regions, fin, g_apply_group.load(std::memory_order_relaxed));
    } else {
        if (many_records) hipLaunchKernelGGL((k_apply_tiles<true, 2>), grid, wg, 0, st, nullptr, sim_bytes, out16, tbase, idx, bin_shift, n_regions, fin, g_apply_group.load(std::memory_order_relaxed));
        else              hipLaunchKernelGGL((k_apply_tiles<false, 2>), grid, wg, 0, st, nullptr, sim_bytes, out16, tbase, idx, bin_shift, n_regions, fin, g_apply_group.load(std::memory_order_relaxed));
    }
}

void launch_apply_by_tiles(uint8_t *sim, size_t sim_bytes, const uint32_t *recs, const uint64_t *binbase, uint32_t n_bins, uint32_t bin_shift,
                           uint32_t *tbase, uint16_t *idx, uint16_t *out16, bool many_records, hipStream_t st, bool big_rows, bool tbase_ready)
{
    launch_sort_tiles(recs, binbase, n_bins, bin_shift, tbase, idx, out16, st, big_rows, tbase_ready);
    const uint32_t n_regions = (uint32_t)((sim_bytes + ((size_t)1 << REGION_SHIFT) - 1) >> REGION_SHIFT);
    const uint32_t grid = apply_tiles_grid(n_regions);
    ApplyFin none; memset(&none, 0, sizeof none);
    // the variant for many records (a step's groups 64 .. 79 in one pass): N = 1e10 (1.2e9 records) 1.09 -> 0.84 ms, configs[4]'s shape (3.2e8)
    // 2.43 -> 2.08; the other one where there are fewer: configs[2] (1.2e8) +3 %, configs[3]'s shape +3 %, text +7 % with the first
    if (many_records) hipLaunchKernelGGL((k_apply_tiles<true, 0>), dim3(grid), dim3(APPLY_WG), 0, st, sim, sim_bytes, out16, tbase, idx, bin_shift, n_regions, none, g_apply_group.load(std::memory_order_relaxed));
    else hipLaunchKernelGGL((k_apply_tiles<false, 0>), dim3(grid), dim3(APPLY_WG), 0, st, sim, sim_bytes, out16, tbase, idx, bin_shift, n_regions, none, g_apply_group.load(std::memory_order_relaxed));
}

// the rows of every 64 KB region of the table (k_apply_tiles, modes 1 and 2): first row, offset of the region's first byte in it, row segments,
// bytes of the region inside the table | (row_off given: none of its rows passed) << 31
__global__ __launch_bounds__(256) void k_region_rows(uint32_t n_regions, uint32_t n_refs, uint64_t table_bytes, const uint64_t *row_off, uint4 *out)
{
    const uint32_t region = blockIdx.x * 256u + threadIdx.x;
    if (region >= n_regions) return;
    const uint64_t rb = (uint64_t)region << REGION_SHIFT;
    const uint32_t len = table_bytes - rb < (1ull << REGION_SHIFT) ? (uint32_t)(table_bytes - rb) : (1u << REGION_SHIFT);
    const uint64_t r0 = rb / n_refs, r1 = (rb + len - 1u) / n_refs;
    const uint32_t skip = row_off && row_off[r1 + 1u] == row_off[r0] ? 1u : 0u;
    out[region] = make_uint4((uint32_t)r0, (uint32_t)(rb - r0 * n_refs), (uint32_t)(r1 - r0) + 1u, len | (skip << 31));
}
void launch_region_rows(uint32_t n_regions, uint32_t n_refs, uint64_t table_bytes, const uint64_t *row_off, void *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_region_rows, dim3((n_regions + 255u) / 256u), dim3(256), 0, st, n_regions, n_refs, table_bytes, row_off, static_cast<uint4 *>(out));
}

// the long clusters' update records bucketed by table region (a few, rarely millions): count, prefix, scatter
__global__ __launch_bounds__(256) void k_bigrec_count(const uint64_t *recs, uint32_t n, uint32_t *cnt)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        atomicAdd(&cnt[(uint32_t)((recs[i] & ((1ull << CELL_BITS) - 1ull)) >> REGION_SHIFT)], 1u);
}
__global__ __launch_bounds__(256) void k_bigrec_scatter(const uint64_t *recs, uint32_t n, const uint64_t *off, uint32_t *cursor, uint64_t *out)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint64_t r = recs[i];
        const uint32_t reg = (uint32_t)((r & ((1ull << CELL_BITS) - 1ull)) >> REGION_SHIFT);
        out[off[reg] + atomicAdd(&cursor[reg], 1u)] = r;
    }
}
void launch_bigrec_buckets(const uint64_t *recs, uint32_t n, uint32_t n_regions, uint32_t *cnt, uint32_t *cursor, uint64_t *off, uint64_t *out, hipStream_t st)
{
    launch_zero2(cnt, (size_t)n_regions * 4u, nullptr, 0, st);
    launch_zero2(cursor, (size_t)n_regions * 4u, nullptr, 0, st);
    const uint32_t grid = n ? ((n + 255u) / 256u < 1024u ? (n + 255u) / 256u : 1024u) : 1u;
    if (n) hipLaunchKernelGGL(k_bigrec_count, dim3(grid), dim3(256), 0, st, recs, n, cnt);
    launch_scan_tiles(cnt, off, n_regions, reinterpret_cast<unsigned long long *>(off + n_regions), st);
    if (n) hipLaunchKernelGGL(k_bigrec_scatter, dim3(grid), dim3(256), 0, st, recs, n, off, cursor, out);
}

uint64_t tiles_bound(uint64_t n_records, uint32_t n_bins) { return n_records / PART_TILE + n_bins; }

uint32_t part_tile() { return PART_TILE; }
uint32_t row_stride() { return ROW_STRIDE; }

void launch_regroup(const uint32_t *rx, const uint64_t *srcoff, uint32_t n_src, uint32_t nb, const uint64_t *dstbase, uint32_t *dst, hipStream_t st)
{
    if (nb) hipLaunchKernelGGL(k_regroup, dim3(nb), dim3(256), 0, st, rx, srcoff, n_src, nb, dstbase, dst);
}

void launch_apply_bigrecs(const uint64_t *recs, uint64_t n, uint64_t cell_lo, uint64_t cell_hi, uint8_t *block, hipStream_t st)
{
    if (!n) return;
    const uint64_t want = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_apply_bigrecs, dim3((uint32_t)(want < 4096u ? want : 4096u)), dim3(256), 0, st, recs, n, cell_lo, cell_hi, block);
}

void launch_tile(int ebwt, int mode, const ScanArgs &a, uint32_t max_blocks, hipStream_t st)
{
    if (mode != 0) launch_scan_kernel<2, ScanCfg<0, 0>::wg>(k_scan<0, 1, 0>, a, max_blocks, st);
    else if (a.upd_mode && a.n_sub <= 2u && !a.no_direct) {                 // the scorers write finished records (k_scan<., 0, 2>)
        if (ebwt) launch_scan_kernel<6, ScanCfg<1, 2>::wg>(k_scan<1, 0, 2>, a, max_blocks, st);
        else      launch_scan_kernel<5, ScanCfg<0, 2>::wg>(k_scan<0, 0, 2>, a, max_blocks, st);
    } else if (a.upd_mode) {
        if (ebwt) launch_scan_kernel<4, ScanCfg<1, 1>::wg>(k_scan<1, 0, 1>, a, max_blocks, st);
        else      launch_scan_kernel<3, ScanCfg<0, 1>::wg>(k_scan<0, 0, 1>, a, max_blocks, st);
    } else {
        if (ebwt) launch_scan_kernel<1, ScanCfg<1, 0>::wg>(k_scan<1, 0, 0>, a, max_blocks, st);
        else      launch_scan_kernel<0, ScanCfg<0, 0>::wg>(k_scan<0, 0, 0>, a, max_blocks, st);
    }
}

void launch_emit(const ScanArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_emit, dim3((a.n_tiles + 3u) / 4u), dim3(256), 0, st, a);
}

void launch_resolve(int mode, const ScanArgs &a, hipStream_t st)
{
    const dim3 grid((a.n_tiles + 255) / 256), block(256);
    if (mode == 0) hipLaunchKernelGGL(k_resolve_open, dim3(64), block, 0, st, a);     // a wave per noted open segment (rare), 256 waves
    else           hipLaunchKernelGGL((k_resolve<1>), grid, block, 0, st, a);
}

void launch_scan_tiles(const uint32_t *cnt, uint64_t *off, uint32_t n, unsigned long long *total, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, cnt, off, n, total);
}

void launch_score_list(int ebwt, const ScanArgs &a, const lime_cluster_t *list, uint64_t count, uint32_t blocks, hipStream_t st)
{
    if (a.upd_mode) {
        if (ebwt) hipLaunchKernelGGL((k_score_list<1, 1>), dim3(blocks), dim3(SCAN_WG), 0, st, a, list, count);
        else      hipLaunchKernelGGL((k_score_list<0, 1>), dim3(blocks), dim3(SCAN_WG), 0, st, a, list, count);
    } else {
        if (ebwt) hipLaunchKernelGGL((k_score_list<1, 0>), dim3(blocks), dim3(SCAN_WG), 0, st, a, list, count);
        else      hipLaunchKernelGGL((k_score_list<0, 0>), dim3(blocks), dim3(SCAN_WG), 0, st, a, list, count);
    }
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

void launch_gather_pairs(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs, const uint64_t *row_off,
                         lime_pair_t *pairs, hipStream_t st)
{
    const uint64_t want = ((uint64_t)n_reads + WGSZ / 64 - 1) / (WGSZ / 64);
    const uint32_t blocks = (uint32_t)(want < 16384u ? (want ? want : 1u) : 16384u);
    hipLaunchKernelGGL(k_gather_pairs, dim3(blocks), dim3(WGSZ), 0, st, sim, n_reads, n_refs, row_off, pairs);
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

__global__ void k_add_u64(uint64_t *p, size_t n, uint64_t v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += v;
}
void launch_add_u64(uint64_t *p, size_t n, uint64_t v, hipStream_t st)
{
    hipLaunchKernelGGL(k_add_u64, dim3(8), dim3(256), 0, st, p, n, v);
}

void launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t st, uint32_t count, size_t pitch)
{
    hipLaunchKernelGGL(k_fill_u32, dim3(count > 1 ? 64 : 1024, count), dim3(256), 0, st, p, n, v, pitch);
}

// zero a few words (the pass's counters) and, with them, a 16-byte-aligned array (the table, where the updates add to it): ONE launch.
// hipMemsetAsync is a kernel as well, but every call of it sat 10 us behind the kernel before it and 6 us in front of the next one
// in the traces (configs[1]: two calls, 30 of the pass's 264 us).
__global__ void k_zero2(uint32_t *a, uint32_t a_words, uint4 *b, size_t b_quads)
{
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < a_words; i += blockDim.x) a[i] = 0u;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < b_quads; i += stride) b[i] = make_uint4(0u, 0u, 0u, 0u);
}

void launch_zero2(void *a, size_t a_bytes, void *b, size_t b_bytes, hipStream_t st)
{
    const size_t quads = b_bytes / 16u;
    const uint32_t grid = quads ? (uint32_t)((quads + 255u) / 256u < 2048u ? (quads + 255u) / 256u : 2048u) : 1u;
    hipLaunchKernelGGL(k_zero2, dim3(grid), dim3(256), 0, st, static_cast<uint32_t *>(a), (uint32_t)(a_bytes / 4u), static_cast<uint4 *>(b), quads);
}

} // namespace lime
