// lime_device.h -- shared device/host pure functions of the LiME hot path (gfx950).
//
// Everything here is arithmetic that must be bit-exact with the reference:
//   sym_index   : src/ClusterBWT_DA.cpp:455-470 (umapIUPAC; unknown bytes -> 0)
//   iupac_match : the value of the pair score when read and genome each occur once
//   pair_score  : src/ClusterBWT_DA.cpp:129-177 on packed 16 x u8 histograms
// They are __host__ __device__ so that the same source is unit-tested on the CPU through
// lime_sym_index / lime_pair_score (include/lime_hip.h) without a GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lime {

#define LIME_HD __host__ __device__ __forceinline__

// byte -> index: A0 C1 G2 T3 R4 Y5 S6 W7 K8 M9 B10 D11 H12 V13 N14, '\0' 15, any other 0.
// Two nibble tables indexed by (byte - 'A'): letters A..P and Q..Z.
LIME_HD uint32_t sym_index(uint32_t b)
{
    const uint64_t LO = 0x00E90800C200B1A0ull;   // P O N M L K J I H G F E D C B A
    const uint64_t HI = 0x0000000507D03640ull;   //             Z Y X W V U T S R Q
    uint32_t c = b - (uint32_t)'A';
    uint64_t tab = (c < 16u) ? LO : HI;
    uint32_t v = (uint32_t)(tab >> ((c & 15u) * 4u)) & 15u;
    v = (c < 26u) ? v : 0u;
    return (b == 0u) ? 15u : v;
}

// nibble a = set of bases {A,C,G,T} (bits 0..3) that IUPAC index a stands for
// (src/ClusterBWT_DA.cpp:472-487); index 15 ('\0') stands for none.
constexpr uint64_t CORR_PACKED = 0x0F7BDE3C96A58421ull;
LIME_HD uint32_t corr_set(uint32_t a) { return (uint32_t)(CORR_PACKED >> (a * 4u)) & 15u; }

// Score of a pair in which the read and the genome each contribute ONE symbol (a, b):
// 1 iff a == b, or one is a base and the other an IUPAC code (4..14) containing it.
// (Derivation from :133-177 with one-hot histograms: DESIGN.md section "pair score".)
LIME_HD uint32_t iupac_match(uint32_t a, uint32_t b)
{
    uint32_t ca = corr_set(a), cb = corr_set(b);
    uint32_t m = (a == b);
    m |= (a < 4u) & (b >= 4u) & ((cb >> a) & 1u);
    m |= (b < 4u) & (a >= 4u) & ((ca >> b) & 1u);
    return m;
}

LIME_HD uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t acc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u8(a, b, acc);
#else
    for (int k = 0; k < 4; k++) {
        int x = (a >> (8 * k)) & 255, y = (b >> (8 * k)) & 255;
        acc += (uint32_t)(x > y ? x - y : y - x);
    }
    return acc;
#endif
}

LIME_HD uint32_t byte_of(const uint32_t (&w)[4], int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 255u; }

// Pair score on packed histograms: byte i of (w[i/4]) = count of symbol index i.
// cr: read counts (already reduced mod 256), cg: genome counts (already saturated at 255).
// Returns t mod 256.
LIME_HD uint32_t pair_score_iupac(const uint32_t (&cr)[4], const uint32_t (&cg)[4]);

LIME_HD uint32_t pair_score(const uint32_t (&cr)[4], const uint32_t (&cg)[4])
{
    // No IUPAC code (indices 4..14) on either side: the cross-match block (:146-177) adds
    // nothing, t = sum_i min(cr_i, cg_i) = (sum cr + sum cg - sum |cr - cg|) / 2.
    uint32_t iu = cr[1] | cr[2] | (cr[3] & 0x00FFFFFFu) | cg[1] | cg[2] | (cg[3] & 0x00FFFFFFu);
    if (iu == 0u) {
        uint32_t sr = sad_u8(cr[0], 0u, sad_u8(cr[3], 0u, 0u));
        uint32_t sg = sad_u8(cg[0], 0u, sad_u8(cg[3], 0u, 0u));
        uint32_t sd = sad_u8(cr[0], cg[0], sad_u8(cr[3], cg[3], 0u));
        return ((sr + sg - sd) >> 1) & 255u;
    }
    return pair_score_iupac(cr, cg);
}

// byte i (0..15) of a packed histogram, dynamic i, no memory indexing
LIME_HD uint32_t hist_get(const uint32_t (&w)[4], uint32_t i)
{
    const uint32_t x = (i & 8u) ? ((i & 4u) ? w[3] : w[2]) : ((i & 4u) ? w[1] : w[0]);
    return (x >> ((i & 3u) * 8u)) & 255u;
}
LIME_HD void hist_set(uint32_t (&w)[4], uint32_t i, uint32_t v)
{
    const uint32_t sh = (i & 3u) * 8u, keep = ~(255u << sh), k = i >> 2;
    w[0] = (k == 0u) ? ((w[0] & keep) | (v << sh)) : w[0];
    w[1] = (k == 1u) ? ((w[1] & keep) | (v << sh)) : w[1];
    w[2] = (k == 2u) ? ((w[2] & keep) | (v << sh)) : w[2];
    w[3] = (k == 3u) ? ((w[3] & keep) | (v << sh)) : w[3];
}

// the general case (rare).  t starts from the sum of minima (as above); the cross-match block
// (:146-177) only ever touches the leftovers of the four bases and of IUPAC codes whose read and
// genome counts differ, and an (i, a) step is a no-op when both leftovers of code a are zero --
// leftovers of codes never grow -- so it is enough to walk, for each base i in order, the codes a
// (ascending) that start with a non-zero leftover.  Same order of effects as the reference.
// The leftovers live in two 64-bit halves (bins 0..7, 8..15) read and written with shifts: plain
// scalars, so that the compiler has no array to move into scratch memory.
LIME_HD uint32_t h64_get(uint64_t lo, uint64_t hi, uint32_t i)
{
    return (uint32_t)(((i & 8u) ? hi : lo) >> ((i & 7u) * 8u)) & 255u;
}
LIME_HD void h64_set(uint64_t &lo, uint64_t &hi, uint32_t i, uint32_t v)
{
    const uint32_t sh = (i & 7u) * 8u;
    const uint64_t keep = ~(255ull << sh), put = (uint64_t)v << sh;
    if (i & 8u) hi = (hi & keep) | put; else lo = (lo & keep) | put;
}

LIME_HD uint32_t pair_score_iupac(const uint32_t (&cr)[4], const uint32_t (&cg)[4])
{
    uint32_t t = (sad_u8(cr[0], 0u, sad_u8(cr[1], 0u, sad_u8(cr[2], 0u, sad_u8(cr[3], 0u, 0u)))) +
                  sad_u8(cg[0], 0u, sad_u8(cg[1], 0u, sad_u8(cg[2], 0u, sad_u8(cg[3], 0u, 0u)))) -
                  sad_u8(cr[0], cg[0], sad_u8(cr[1], cg[1], sad_u8(cr[2], cg[2], sad_u8(cr[3], cg[3], 0u))))) >> 1;
    const uint64_t crl = cr[0] | ((uint64_t)cr[1] << 32), crh = cr[2] | ((uint64_t)cr[3] << 32);
    const uint64_t cgl = cg[0] | ((uint64_t)cg[1] << 32), cgh = cg[2] | ((uint64_t)cg[3] << 32);
    uint64_t rrl = 0, rrh = 0, rgl = 0, rgh = 0;                   // leftovers, packed like the histograms
    uint32_t nz = 0;                                               // codes 4..14 with a non-zero leftover
    for (uint32_t i = 0; i < 15u; i++) {
        const uint32_t a = h64_get(crl, crh, i), b = h64_get(cgl, cgh, i);
        if (i >= 4u && a == b) continue;
        const uint32_t mn = a < b ? a : b;
        h64_set(rrl, rrh, i, a - mn); h64_set(rgl, rgh, i, b - mn);
        if (i >= 4u) nz |= 1u << i;
    }
    for (uint32_t i = 0; i < 4u; i++) {          // :146-177
        uint32_t m = nz;
        while (m) {
            const uint32_t a = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            if (!((CORR_PACKED >> (a * 4u + i)) & 1ull)) continue;
            const uint32_t ga = h64_get(rgl, rgh, a), ri = h64_get(rrl, rrh, i);
            if (ga > 0u) {                       // :150-161 (as written: the zeroed side is "subtracted")
                if (ga > ri) { t += ri; h64_set(rrl, rrh, i, 0u); }
                else         { t += ga; h64_set(rgl, rgh, a, 0u); }
            }
            const uint32_t ra = h64_get(rrl, rrh, a), gi = h64_get(rgl, rgh, i);
            if (ra > 0u) {                       // :163-174
                if (ra > gi) { t += gi; h64_set(rrl, rrh, a, ra - gi); h64_set(rgl, rgh, i, 0u); }
                else         { t += ra; h64_set(rgl, rgh, i, gi - ra); h64_set(rrl, rrh, a, 0u); }
            }
        }
    }
    return t & 255u;
}

// histogram increment of symbol index s (0..15) without dynamic register indexing
LIME_HD void hist_add(uint32_t (&w)[4], uint32_t s, uint32_t on)
{
    uint32_t inc = on << ((s & 3u) * 8u);
    uint32_t k = s >> 2;
    w[0] += (k == 0u) ? inc : 0u;
    w[1] += (k == 1u) ? inc : 0u;
    w[2] += (k == 2u) ? inc : 0u;
    w[3] += (k == 3u) ? inc : 0u;
}

// splitmix64 finaliser; synthetic inputs of SURVEY.md 8(d) (identical to oracle/lime_oracle.c)
LIME_HD uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}

LIME_HD void synth_element(uint64_t seed, uint64_t i, uint32_t n_reads, uint32_t n_refs,
                           uint32_t alpha, uint32_t mode, uint32_t &l, uint32_t &d, uint32_t &s)
{
    const uint64_t PHI = 0x9E3779B97F4A7C15ull;
    const uint32_t p_run = mode ? 24904u : 26214u;
    const uint32_t p_read = mode ? 16384u : 6554u;
    uint64_t a = mix64(seed ^ (i * PHI));
    uint64_t b = mix64(a + PHI);
    uint32_t u = (uint32_t)(a & 0xFFFF), r1 = (uint32_t)((a >> 16) & 0xFFFF);
    uint32_t v = (uint32_t)((a >> 32) & 0xFFFF);
    l = (u < p_run) ? alpha + (r1 % 48u) : (alpha ? r1 % alpha : 0u);
    if (i == 0) l = 0;
    uint32_t sel = (uint32_t)(b & 0xFFFFFFFFu);
    d = (v < p_read) ? (sel % n_reads) : n_reads + (sel % n_refs);
    uint32_t w = (uint32_t)((b >> 32) & 0xFFFF), q = (uint32_t)(b >> 48);
    const uint32_t ACGT = 0x54474341u;           // 'A','C','G','T' little-endian
    if (w < 63570u) {
        uint32_t k = q & 3u;
        if (mode && (q >> 2) % 10u != 0u) {
            uint64_t c = mix64(seed ^ ((i >> 3) * PHI) ^ 0xA5A5A5A5A5A5A5A5ull);
            k = (uint32_t)(c & 3u);
        }
        s = (ACGT >> (k * 8u)) & 255u;
    } else if (w < 64225u) s = 'N';
    else if (w < 64881u) s = 0;
    else s = (q & 1u) ? 'Y' : 'R';
}

} // namespace lime
