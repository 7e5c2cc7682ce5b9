"""End to end, the way LiME_paired.sh runs it (LiME_paired.sh:44-81): four collections (reads_1, its reverse
complement, reads_2, its reverse complement, each with the genomes) -> ClusterLCP -> ClusterBWT_DA -> Classify.
The drop-in programs (GPU for the two cluster steps) against the reference's own programs (oracle/_ref, CPU):
every intermediate and final file byte-identical.  Needs oracle/_ref (built from /root/reference by
`make -C oracle`; the binaries travel to the GPU box with the snapshot) -- skipped without it."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "lime_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _collection(seed):
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = [bytes(rng.choice(bases, 260)) for _ in range(8)]
    genomes[5] = genomes[4][:200] + genomes[5][200:]            # two close relatives
    genomes[2] = genomes[2][:100] + b"N" + genomes[2][101:]     # an IUPAC code in a genome
    r1, r2 = [], []
    for _ in range(30):
        g = genomes[rng.integers(0, 8)]
        s = int(rng.integers(0, 260 - 90))
        frag = bytearray(g[s:s + 90])
        for k in rng.integers(0, 90, 2):
            frag[k] = int(bases[rng.integers(0, 4)])            # sequencing errors
        r1.append(bytes(frag[:40]))
        r2.append(bytes(frag[50:90]).translate(COMP)[::-1])     # the mate, from the other strand
    r1 += [bytes(rng.choice(bases, 40)) for _ in range(4)]      # reads from nowhere
    r2 += [bytes(rng.choice(bases, 40)) for _ in range(4)]
    return genomes, r1, r2


def _rc(reads):
    return [r.translate(COMP)[::-1] for r in reads]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Classify")), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [1, 2])
def test_pipeline_matches_reference_chain(tmp_path, seed):
    from lime_amd.builder import build_arrays
    genomes, r1, r2 = _collection(seed)
    n_reads, n_refs, alpha, read_len, beta = len(r1), len(genomes), 16, 40, 0.25
    sets = {"F1": r1, "F1RC": _rc(r1), "F2": r2, "F2RC": _rc(r2)}
    tax = tmp_path / "lineage.csv"
    rows = ["Accession_number;Species_TaxID;Genus_TaxID;Family_TaxID;Order_TaxID;Class_TaxID;Phylum_TaxID"]
    rows += [f"ACC{g}.1;{100 + g // 2};{200 + g // 4};300;400;500;600" for g in range(n_refs)]
    tax.write_bytes(("\n".join(rows) + "\n").encode())

    def chain(bindir, tag, env=None):
        d = tmp_path / tag
        d.mkdir()
        res = []
        for name, reads in sets.items():
            base = str(d / f"{name}.fasta")
            ebwt, lcp, da = build_arrays(reads, genomes)
            lcp.astype("<u4").tofile(base + ".lcp"); da.astype("<u4").tofile(base + ".da"); ebwt.tofile(base + ".ebwt")
            subprocess.run([os.path.join(bindir, "ClusterLCP"), base, str(n_reads), str(n_refs), str(alpha), "1"],
                           check=True, capture_output=True, cwd=d, timeout=120, env=env)
            subprocess.run([os.path.join(bindir, "ClusterBWT_DA"), base, str(read_len), str(beta), "1"],
                           check=True, capture_output=True, cwd=d, timeout=120, env=env)
            res.append(base + ".res")
        out = str(d / "classification.txt")
        # the four result files in the script's order F1, F1RC, F2, F2RC (Classify adds files 0+3 and 1+2)
        subprocess.run([os.path.join(bindir, "Classify"), "4"] + res + [str(n_reads), str(n_refs), out, str(tax), "1", "1"],
                       check=True, capture_output=True, cwd=d, timeout=120, env=env)
        files = {}
        for f in sorted(os.listdir(d)):
            files[f] = open(d / f, "rb").read()
        return files

    ref = chain(REF, "ref")
    ours = chain(OURS, "ours")
    assert sorted(ref) == sorted(ours)
    for f in ref:
        assert ref[f] == ours[f], f
    lines = ours["classification.txt"].decode().splitlines()[1:]
    assert len(lines) == n_reads and sum(1 for l in lines if l[0] == "C") > n_reads // 2
