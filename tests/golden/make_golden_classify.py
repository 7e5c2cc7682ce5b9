#!/usr/bin/env python3
"""Golden vectors for the read-assignment step, made with the REFERENCE'S OWN Classify
binaries (oracle/_ref/Classify{,_txt,_higher,_higher_txt}, built by `make -C oracle`).

    python tests/golden/make_golden_classify.py

Each case classify_<name>.npz holds: the similarity tables of the 2 or 4 .res inputs (u8, the
files themselves are rebuilt from them with the oracle's writers, which tests/test_oracle_golden.py
pins to the reference's ClusterBWT_DA byte for byte), norm, beta, the taxonomy file's bytes, and
for every (binary, higher, rank) combination the bytes of the classification file the reference
wrote.  Data only; nothing of the reference's source is stored.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

RANKS = (0, 1, 2, 4)


def taxonomy(n_targ, rng, holes):
    """lineage file: genomes share species / genus / family in groups; `holes` leaves some fields empty
    at the higher ranks (never at species/genus so that every rank tested keeps one taxon per genome)"""
    lines = ["Accession_number;Species_TaxID;Genus_TaxID;Family_TaxID;Order_TaxID;Class_TaxID;Phylum_TaxID"]
    for g in range(n_targ):
        sp = 100 + g // 2
        ge = 200 + g // 4
        fa = 300 + g // 6
        od = 400 + g // 8
        cl = 500 + g // 12
        ph = 600
        f = [f"ACC_{g:03d}.1", str(sp), str(ge), str(fa), str(od), str(cl), str(ph)]
        if holes and rng.random() < 0.2:
            f[3] = ""
        if holes and rng.random() < 0.1:
            f[5] = ""
        lines.append(";".join(f))
    return ("\n".join(lines) + "\n").encode()


def tables(n_files, n_reads, n_targ, rng, norm):
    """similarity tables with many near-ties: a read's true genome gets a high count in the forward
    file(s), relatives get counts within a few units, mates and reverse strands are noisy copies"""
    base = np.zeros((n_reads, n_targ), dtype=np.int64)
    for r in range(n_reads):
        kind = rng.random()
        if kind < 0.12:
            continue                                   # read without any similarity
        g = rng.integers(0, n_targ)
        top = int(rng.integers(int(0.2 * norm), norm))
        base[r, g] = top
        for _ in range(rng.integers(0, 4)):            # relatives: same / close values
            h = (g + rng.integers(-3, 4)) % n_targ
            base[r, h] = max(0, top - int(rng.integers(0, 4)))
        for _ in range(rng.integers(0, 3)):            # unrelated low hits
            base[r, rng.integers(0, n_targ)] = int(rng.integers(1, max(2, top // 2)))
    out = []
    for i in range(n_files):
        t = base.copy()
        noise = rng.integers(-2, 3, size=t.shape)
        t = np.where(t > 0, np.clip(t + noise * (rng.random(t.shape) < 0.5), 0, 255), 0)
        if i % 2 == 1:                                 # the other strand: mostly weaker, sometimes equal
            t = np.where(rng.random(t.shape) < 0.6, t // 3, t)
        drop = rng.random(n_reads) < 0.15
        t[drop] = 0
        out.append(t.astype(np.uint8))
    return out


def run_reference(sims, norm, beta, tax_bytes, n_targ):
    res = {}
    n_files, n_reads = len(sims), sims[0].shape[0]
    with tempfile.TemporaryDirectory() as td:
        bases = []
        for i, s in enumerate(sims):
            b = os.path.join(td, f"in{i}.res")
            O.write_res_txt(b + ".txt", s, norm, beta)
            O.write_res_bin(b + ".bin", b + ".pos", s, norm, beta)
            bases.append(b)
        tax = os.path.join(td, "lineage.csv")
        open(tax, "wb").write(tax_bytes)
        for binary in (1, 0):
            for higher in (0, 1):
                exe = os.path.join(REF, "Classify" + ("_higher" if higher else "") + ("" if binary else "_txt"))
                for rank in RANKS:
                    if higher and rank == 0:
                        continue                       # the reference indexes rank-1 = -1 there
                    outp = os.path.join(td, f"o_{binary}{higher}{rank}.txt")
                    cmd = [exe, str(n_files)] + bases + [str(n_reads), str(n_targ), outp, tax, str(rank), "1"]
                    p = subprocess.run(cmd, cwd=td, capture_output=True, timeout=120)
                    if p.returncode != 0:
                        raise RuntimeError(f"{cmd}: {p.stderr.decode()[-300:]}")
                    res[f"out_b{binary}_h{higher}_r{rank}"] = np.frombuffer(open(outp, "rb").read(), dtype=np.uint8)
    return res


def main():
    cases = {
        "single": dict(n_files=2, n_reads=300, n_targ=12, seed=1, holes=False, norm=85, beta=0.25),
        "paired": dict(n_files=4, n_reads=400, n_targ=12, seed=2, holes=False, norm=85, beta=0.25),
        "paired_holes": dict(n_files=4, n_reads=300, n_targ=24, seed=3, holes=True, norm=60, beta=0.1),
        "single_tiny": dict(n_files=2, n_reads=3, n_targ=2, seed=4, holes=False, norm=10, beta=0.0),
    }
    for name, c in cases.items():
        rng = np.random.default_rng(c["seed"])
        tax = taxonomy(c["n_targ"], rng, c["holes"])
        sims = tables(c["n_files"], c["n_reads"], c["n_targ"], rng, c["norm"])
        ref = run_reference(sims, c["norm"], c["beta"], tax, c["n_targ"])
        path = os.path.join(HERE, f"classify_{name}.npz")
        np.savez_compressed(path, sims=np.stack(sims), norm=c["norm"], beta=c["beta"],
                            tax=np.frombuffer(tax, dtype=np.uint8), **ref)
        kinds = {}
        for k, v in ref.items():
            for line in v.tobytes().decode().splitlines()[1:]:
                kinds.setdefault(k, {}).setdefault(line[0], 0)
                kinds[k][line[0]] += 1
        print(name, os.path.getsize(path), "bytes;", {k: kinds[k] for k in list(kinds)[:4]})


if __name__ == "__main__":
    main()
