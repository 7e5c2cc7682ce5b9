#!/usr/bin/env python3
"""Golden vector standing in for BASELINE.json configs[0] AT THE EXAMPLE'S FULL SIZE (README.md:125-131, LiME_paired.sh:44-79):
all 20 000 example reads (example/reads_1.fasta + reads_2.fasta, 10 000 pairs x 100 bp from four source genomes named in their
headers), paired-end, four collections (reads_1, its reverse complement, reads_2, its reverse complement -- each with the genomes)
through ClusterLCP -> ClusterBWT_DA -> Classify 4.  example/refs.fasta is absent and the suffix-array tools need the network, so the
genomes are SURROGATES built from the reads: for each of the three accessions of example/LineageFile.csv, its 2 500 fragments laid
end to end, a fragment = read_1 followed by the reverse complement of read_2 (so reads_1 and rc(reads_2) occur in it, their mates'
strands do not -- as with a real forward-strand reference); CP000360 has no genome (negative control, like in the example).
1 % of the reads' bases are substituted afterwards (seeded), so that runs break the way sequencing errors break them.

Stored (tests/golden/example_full.npz): the two read sets (data of the reference's example, 2 bits per base after compression),
their source index, the taxonomy rows, and the REFERENCE'S OWN outputs (oracle/_ref, 1 thread): sha256 of every intermediate file of
the four collections and the full classification file.  No text of the reference.  Run in the build container:
  python tests/golden/make_golden_example.py"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lime_amd.builder import build_arrays_sa  # noqa: E402

EX = "/root/reference/example"
DB = ["AP009048", "CP001845", "CR543861"]          # example/LineageFile.csv:2-4
ALPHA, READ_LEN, BETA = 16, 100, 0.25
COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
SETS = ("F1", "F1RC", "F2", "F2RC")


def rc(b):
    return bytes(b).translate(COMP)[::-1]


def read_fasta(path):
    names, seqs = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:].split("-")[0])
            elif line:
                seqs.append(line.upper().encode())
    return names, seqs


def collections(r1, r2, src):
    """-> (genomes, {set name: reads}) exactly as the GPU test rebuilds them from the fixture (r1, r2: uint8 [n][100])."""
    r1 = [bytes(x) for x in r1]; r2 = [bytes(x) for x in r2]
    genomes = []
    for g in range(len(DB)):
        idx = [i for i in range(len(r1)) if src[i] == g]
        genomes.append(b"".join(r1[i] + rc(r2[i]) for i in idx))
    rng = np.random.default_rng(2026)
    def noisy(reads):
        out = []
        for r in reads:
            r = bytearray(r)
            for p in np.nonzero(rng.random(len(r)) < 0.01)[0]:
                r[p] = b"ACGT"[rng.integers(0, 4)]
            out.append(bytes(r))
        return out
    n1, n2 = noisy(r1), noisy(r2)
    return genomes, {"F1": n1, "F1RC": [rc(x) for x in n1], "F2": n2, "F2RC": [rc(x) for x in n2]}


def run_chain(bindir, d, genomes, sets, n_reads, lineage_bytes, concurrent=False):
    """LiME_paired.sh:44-79 with the programs of `bindir`; returns {file name: bytes} of everything it wrote."""
    n_refs = len(genomes)
    bases = {}
    for name in SETS:
        base = os.path.join(d, f"{name}.fasta")
        ebwt, lcp, da = build_arrays_sa(sets[name], genomes, term=0)
        lcp.astype("<u4").tofile(base + ".lcp"); da.astype("<u4").tofile(base + ".da"); ebwt.tofile(base + ".ebwt")
        bases[name] = base
    procs = []
    for name in SETS:                               # step 1: the script starts the four ClusterLCP together
        cmd = [os.path.join(bindir, "ClusterLCP"), bases[name], str(n_reads), str(n_refs), str(ALPHA), "1"]
        if concurrent:
            procs.append(subprocess.Popen(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        else:
            subprocess.run(cmd, check=True, capture_output=True, cwd=d, timeout=600)
    for p in procs:
        assert p.wait(timeout=600) == 0
    for name in SETS:                               # step 2: one ClusterBWT_DA after the other
        subprocess.run([os.path.join(bindir, "ClusterBWT_DA"), bases[name], str(READ_LEN), str(BETA), "1"],
                       check=True, capture_output=True, cwd=d, timeout=600)
    tax = os.path.join(d, "LineageFile.csv")
    open(tax, "wb").write(lineage_bytes)
    out = os.path.join(d, "classification.txt")
    subprocess.run([os.path.join(bindir, "Classify"), "4"] + [bases[n] + ".res" for n in SETS] +
                   [str(n_reads), str(n_refs), out, tax, "1", "1"], check=True, capture_output=True, cwd=d, timeout=600)
    files = {}
    for f in sorted(os.listdir(d)):
        if f.endswith((".lcp", ".da", ".ebwt")) or f == "LineageFile.csv":
            continue
        files[f] = open(os.path.join(d, f), "rb").read()
    return files


def main():
    n1, s1 = read_fasta(os.path.join(EX, "reads_1.fasta"))
    n2, s2 = read_fasta(os.path.join(EX, "reads_2.fasta"))
    assert n1 == n2 and len(s1) == len(s2) == 10000 and all(len(x) == 100 for x in s1 + s2)
    src = np.array([DB.index(a) if a in DB else -1 for a in n1], dtype=np.int8)
    r1 = np.frombuffer(b"".join(s1), dtype=np.uint8).reshape(-1, 100)
    r2 = np.frombuffer(b"".join(s2), dtype=np.uint8).reshape(-1, 100)
    lineage = open(os.path.join(EX, "LineageFile.csv"), "rb").read()
    genomes, sets = collections(r1, r2, src)
    with tempfile.TemporaryDirectory() as d:
        files = run_chain(os.path.join(ROOT, "oracle", "_ref"), d, genomes, sets, len(r1), lineage)
    names = sorted(files)
    cls = files["classification.txt"]
    lines = cls.decode().splitlines()
    print(f"example_full: {len(r1)} pairs, genomes of {[len(g) for g in genomes]} bases; files {names}")
    print("classification head:", lines[:3], "...", len(lines), "lines")
    np.savez_compressed(os.path.join(HERE, "example_full.npz"), reads_1=r1, reads_2=r2, src=src,
                        lineage=np.frombuffer(lineage, dtype=np.uint8),
                        file_names=np.array(names), file_sha256=np.array([hashlib.sha256(files[f]).hexdigest() for f in names]),
                        file_sizes=np.array([len(files[f]) for f in names], dtype=np.int64),
                        classification=np.frombuffer(cls, dtype=np.uint8))


if __name__ == "__main__":
    main()
