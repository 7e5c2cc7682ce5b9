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


def suffix_array_doubling(keys):
    """Suffix array of an integer sequence by prefix doubling (Manber-Myers with numpy sorts): O(n log n) per round,
    ceil(log2(longest repeat)) rounds.  keys: int64 array; returns the suffix order (int64[n])."""
    keys = np.asarray(keys, dtype=np.int64)
    n = len(keys)
    order = np.argsort(keys, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    sk = keys[order]
    rank[order] = np.concatenate(([0], np.cumsum(sk[1:] != sk[:-1])))
    k = 1
    while k < n and rank.max() < n - 1:
        second = np.full(n, -1, dtype=np.int64)
        second[:n - k] = rank[k:]
        order = np.lexsort((second, rank))
        r1, r2 = rank[order], second[order]
        rank[order] = np.concatenate(([0], np.cumsum((r1[1:] != r1[:-1]) | (r2[1:] != r2[:-1]))))
        k *= 2
    return np.argsort(rank, kind="stable")


def build_arrays_sa(reads, genomes, term: int = 0):
    """Same contract as build_arrays (ebwt u8[N], lcp u32[N], da u32[N] of the collection reads + genomes) for
    collections of 10^5..10^6 symbols: generalized suffix array by prefix doubling over the concatenation with one
    distinct terminator per document (terminators sort before every base and among themselves by document id, so no
    comparison runs past a document's end), LCP by Kasai's algorithm.  The role BCR_LCP_GSA / eGSA / eGap play for
    the reference (Preprocessing.sh:156-173); fixtures and demos only."""
    docs = [np.frombuffer(d.encode() if isinstance(d, str) else bytes(d), dtype=np.uint8) for d in list(reads) + list(genomes)]
    nd = len(docs)
    lens = np.array([len(d) + 1 for d in docs], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    n = int(lens.sum())
    keys = np.empty(n, dtype=np.int64)
    text = np.empty(n, dtype=np.uint8)                  # the symbols, terminators written as `term`
    doc_of = np.empty(n, dtype=np.uint32)
    for k, d in enumerate(docs):
        s = int(starts[k])
        keys[s:s + len(d)] = nd + d.astype(np.int64)
        keys[s + len(d)] = k
        text[s:s + len(d)] = d
        text[s + len(d)] = term
        doc_of[s:s + len(d) + 1] = k
    sa = suffix_array_doubling(keys)
    da = doc_of[sa]
    is_start = np.zeros(n, dtype=bool); is_start[starts] = True
    prev = np.where(sa > 0, sa - 1, 0)
    # the symbol before a document's first suffix is the document's own terminator (circular), written as `term`
    ebwt = np.where(is_start[sa], np.uint8(term), text[prev]).astype(np.uint8)
    # Kasai: lcp[rank[i]] from lcp[rank[i-1]] - 1; terminators are distinct keys, so matches stop at a document's end
    rank = np.empty(n, dtype=np.int64); rank[sa] = np.arange(n)
    lcp = np.zeros(n, dtype=np.uint32)
    kl = keys.tolist(); sal = sa.tolist(); rl = rank.tolist()
    h = 0
    for i in range(n):
        r = rl[i]
        if r > 0:
            j = sal[r - 1]
            while i + h < n and j + h < n and kl[i + h] == kl[j + h] and kl[i + h] >= nd:
                h += 1
            lcp[r] = h
            if h > 0:
                h -= 1
        else:
            h = 0
    return ebwt, lcp, da.astype(np.uint32)
