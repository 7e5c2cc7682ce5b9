#!/usr/bin/env python3
"""Text-derived golden vector standing in for BASELINE.json configs[0] (README.md:125-131 quick test): the example
reads (/root/reference/example/reads_1.fasta, 10 000 x 100 bp from four source genomes named in their headers) against
SURROGATE genomes -- example/refs.fasta is absent and the reference's suffix-array tools need the network.

Collection: 2 000 reads (500 per source accession) + 3 genomes for the three accessions of example/LineageFile.csv
(CP000360 has none: negative control, like in the example).  A surrogate genome = 800 reads of its accession laid end
to end (the 500 sampled ones among them, so they occur in it exactly), then: a 2 kb region of genome 0 copied into
genome 1 (reads of one source matching two genomes), 1 % of the sampled reads' bases substituted (sequencing errors
break runs), IUPAC codes N / R / Y / K sprinkled over genome 2 (the cross-match block of ClusterBWT_DA.cpp:146-177).
ebwt / lcp / da come from lime_amd.builder.build_arrays_sa (prefix-doubling suffix sort); the expected outputs from the
REFERENCE'S OWN BINARIES (oracle/_ref), exactly like make_golden.py.  Stored: arrays and reference outputs only -- no
text of the reference.  Run in the build container:  python tests/golden/make_golden_text.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from lime_amd.builder import build_arrays_sa  # noqa: E402
from make_golden import reference_outputs  # noqa: E402

FASTA = "/root/reference/example/reads_1.fasta"
DB = ["AP009048", "CP001845", "CR543861"]          # example/LineageFile.csv:2-4
ALPHA, READ_LEN, BETA = 16, 100, 0.25


def main():
    by_src = {}
    with open(FASTA) as f:
        name = None
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split("-")[0]
            elif name:
                by_src.setdefault(name, []).append(line.upper().encode())
    rng = np.random.default_rng(2026)
    reads, genomes, truth = [], [], []
    for src in sorted(by_src):
        pool = by_src[src]
        pick = rng.choice(len(pool), 800, replace=False)
        sampled = pick[:500]
        if src in DB:
            genomes.append(bytearray(b"".join(pool[i] for i in pick)))
        for i in sampled:
            r = bytearray(pool[i])
            for p in np.nonzero(rng.random(len(r)) < 0.01)[0]:
                r[p] = b"ACGT"[rng.integers(0, 4)]
            reads.append(bytes(r)); truth.append(DB.index(src) if src in DB else -1)
    genomes[1][30000:32000] = genomes[0][10000:12000]           # a region shared by two genomes
    g2 = genomes[2]
    for p in rng.choice(len(g2), 400, replace=False):
        g2[p] = b"NRYK"[rng.integers(0, 4)]
    order = rng.permutation(len(reads))                         # reads of the four sources interleaved
    reads = [reads[i] for i in order]; truth = np.array([truth[i] for i in order], dtype=np.int32)
    ebwt, lcp, da = build_arrays_sa(reads, [bytes(g) for g in genomes], term=0)
    n_reads, n_refs = len(reads), len(genomes)
    out = reference_outputs(lcp, da, ebwt, n_reads, n_refs, ALPHA, READ_LEN, BETA)
    sim = out["sim_e1"]
    best = sim.argmax(axis=1)
    hit = (sim.max(axis=1) > 0)
    ok = int(((best == truth) & hit & (truth >= 0)).sum())
    clustered = int(out["clrs"][:, 1].sum())
    print(f"text_example: {len(lcp)} symbols, {n_reads} reads x {n_refs} genomes, {len(out['clrs'])} clusters "
          f"({clustered / len(lcp):.1%} of the symbols, longest {int(out['clrs'][:, 1].max())}), "
          f"{int((sim > 0).sum())} non-zero cells, argmax == source for {ok} of {int((truth >= 0).sum())} reads of DB sources, "
          f"{int(hit[truth < 0].sum())} of {int((truth < 0).sum())} control reads hit something")
    np.savez_compressed(os.path.join(HERE, "text_example.npz"), lcp=lcp.astype(np.uint32), da=da.astype(np.uint32), ebwt=ebwt,
                        params=np.array([n_reads, n_refs, ALPHA, READ_LEN], dtype=np.int64), beta=np.float64(BETA),
                        truth=truth, **out)


if __name__ == "__main__":
    main()
