"""Tiny in-repo builder of ebwt/lcp/da for toy collections.

The reference delegates this to external tools (BCR_LCP_GSA / eGSA / eGap,
/root/reference/Preprocessing.sh:156-173) that cannot be fetched here.  This is a naive
O(n^2 log n) suffix sort for test fixtures and demos only -- NOT on the hot path.

Conventions follow /root/reference/README.md:7-10 and SURVEY.md Appendix B: documents are the
reads (ids 0..numReads-1) followed by the genomes; every document ends with a terminator
that sorts before every base, terminators of different documents compare by document id;
ebwt[i] is the symbol preceding suffix i (the document's own terminator, written as byte
`term`, for the whole-document suffix); lcp[i] = LCP(suffix i-1, suffix i), lcp[0] = 0.
"""
from __future__ import annotations

import numpy as np


def build_arrays(reads, genomes, term: int = 0):
    """reads, genomes: lists of bytes/str.  Returns (ebwt u8[N], lcp u32[N], da u32[N])."""
    docs = [d.encode() if isinstance(d, str) else bytes(d) for d in list(reads) + list(genomes)]
    # Each symbol becomes an integer key: terminator of doc k -> k (all < 1<<20), base b -> (1<<20)+b
    BASE = 1 << 20
    keyed = [[BASE + c for c in d] + [k] for k, d in enumerate(docs)]
    suffixes = []
    for k, seq in enumerate(keyed):
        for p in range(len(seq)):
            suffixes.append((seq[p:], k, p))
    suffixes.sort(key=lambda s: s[0])
    n = len(suffixes)
    ebwt = np.zeros(n, dtype=np.uint8)
    lcp = np.zeros(n, dtype=np.uint32)
    da = np.zeros(n, dtype=np.uint32)
    prev = None
    for i, (seq, k, p) in enumerate(suffixes):
        da[i] = k
        ebwt[i] = docs[k][p - 1] if p > 0 else term
        if prev is not None:
            l = 0
            m = min(len(seq), len(prev))
            # a terminator never matches another document's terminator (distinct keys)
            while l < m and seq[l] == prev[l] and seq[l] >= BASE:
                l += 1
            lcp[i] = l
        prev = seq
    return ebwt, lcp, da
