#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE'S OWN BINARIES.

Run in the build container only (needs oracle/_ref/, built by `make -C oracle` from the
sources under /root/reference):

    python tests/golden/make_golden.py

For every case it writes <case>.npz holding the inputs (lcp, da, ebwt, parameters) and the
bytes the reference programs produced for them:
  clrs      ClusterLCP 1 thread, fileFasta.<alpha>.clrs   (src/ClusterLCP.cpp:229-235)
  out       the 28-byte aux file                           (src/ClusterLCP.cpp:294-310)
  sim_e1/0  integer SimArray_ dump of the -DSMALL=1 build  (src/ClusterBWT_DA.cpp:672-681)
  txt_e1/0  fileFasta.res.txt of the BIN=0 build           (src/ClusterBWT_DA.cpp:414-441)
  bin_e1/0, pos_e1/0  fileFasta.res.bin/.res.pos, BIN=1    (src/ClusterBWT_DA.cpp:376-436)
It also re-runs ClusterLCP with 4 threads and checks the record SET is unchanged.
Nothing of the reference's source is stored: fixtures are data only.
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
from lime_amd.builder import build_arrays  # noqa: E402


def run(cmd, cwd, timeout=120):
    p = subprocess.run(cmd, cwd=cwd, capture_output=True, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"{cmd} failed: {p.stderr.decode()[-400:]}")
    return p.stdout


def parse_small(stdout, n_reads, n_refs):
    txt = stdout.decode()
    body = txt.split("***FINAL***\n", 1)[1].split("***********", 1)[0]
    vals = np.array(body.split(), dtype=np.int64)
    assert vals.size == n_reads * n_refs, (vals.size, n_reads, n_refs)
    return vals.astype(np.uint8).reshape(n_reads, n_refs)


def reference_outputs(lcp, da, ebwt, n_reads, n_refs, alpha, read_len, beta):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        base = os.path.join(td, "X.fasta")
        lcp.astype("<u4").tofile(base + ".lcp")
        da.astype("<u4").tofile(base + ".da")
        ebwt.astype(np.uint8).tofile(base + ".ebwt")
        clrs_path = f"{base}.{alpha}.clrs"
        # 4 threads first: same SET of records (order is nondeterministic, :229-235)
        # (the reference spins forever when a non-last thread's open run reaches EOF,
        #  src/ClusterLCP.cpp:249-262 -- such cases skip the 4-thread cross-check)
        try:
            run([f"{REF}/ClusterLCP", base, str(n_reads), str(n_refs), str(alpha), "4"], td, timeout=20)
            c4 = np.fromfile(clrs_path, dtype="<u8").reshape(-1, 2)
        except subprocess.TimeoutExpired:
            print("  (4-thread ClusterLCP did not terminate on this case: cross-check skipped)")
            c4 = None
        run([f"{REF}/ClusterLCP", base, str(n_reads), str(n_refs), str(alpha), "1"], td)
        c1 = np.fromfile(clrs_path, dtype="<u8").reshape(-1, 2)
        assert np.array_equal(c1[np.argsort(c1[:, 0])], c1), "1-thread order must be sorted"
        assert c4 is None or np.array_equal(c4[np.argsort(c4[:, 0])], c1), "thread-count dependent SET"
        out["clrs"] = c1
        out["out"] = np.fromfile(os.path.join(td, "X.out"), dtype=np.uint8)
        args = [base, str(read_len), repr(float(beta)), "1"]
        for tag, suf in (("e1", ""), ("e0", "_e0")):
            so = run([f"{REF}/ClusterBWT_DA{suf}_small", *args], td)
            out[f"sim_{tag}"] = parse_small(so, n_reads, n_refs)
            out[f"txt_{tag}"] = np.fromfile(base + ".res.txt", dtype=np.uint8)
            os.remove(base + ".res.txt")
            run([f"{REF}/ClusterBWT_DA{suf}_txt", *args], td)
            assert np.array_equal(out[f"txt_{tag}"], np.fromfile(base + ".res.txt", dtype=np.uint8))
            run([f"{REF}/ClusterBWT_DA{suf}", *args], td)
            out[f"bin_{tag}"] = np.fromfile(base + ".res.bin", dtype=np.uint8)
            out[f"pos_{tag}"] = np.fromfile(base + ".res.pos", dtype=np.uint8)
            # 4 threads: same integer table
            so4 = run([f"{REF}/ClusterBWT_DA{suf}_small", base, str(read_len), repr(float(beta)), "4"], td)
            assert np.array_equal(parse_small(so4, n_reads, n_refs), out[f"sim_{tag}"])
    return out


def rand_dna(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tolist())


def case_toy_text(rng):
    """Reads sampled from toy genomes (shared region, IUPAC codes in one genome and in a few
    reads), suffix-sorted by lime_amd.builder: the text-derived stand-in for config C1."""
    shared = rand_dna(rng, 120)
    g0 = rand_dna(rng, 200) + shared + rand_dna(rng, 150)
    g1 = rand_dna(rng, 150) + shared + rand_dna(rng, 200)
    g2 = bytearray(rand_dna(rng, 400))
    for p in rng.choice(len(g2), size=30, replace=False):
        g2[p] = int(rng.choice(np.frombuffer(b"RYSWKMBDHVN", dtype=np.uint8)))
    genomes = [g0, g1, bytes(g2)]
    reads = []
    for k in range(60):
        g = genomes[k % 3]
        s = int(rng.integers(0, len(g) - 40))
        r = bytearray(g[s:s + 40])
        if k % 7 == 0:
            r[int(rng.integers(0, 40))] = ord("N")
        if k % 11 == 0:
            r[int(rng.integers(0, 40))] = ord("R")
        reads.append(bytes(r))
    ebwt, lcp, da = build_arrays(reads, genomes, term=0)
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=60, n_refs=3, alpha=16, read_len=40, beta=0.25)


def iid(rng, n, n_reads, n_refs, alpha, p_run, p_read, symbols, sym_p=None):
    hi = rng.random(n) < p_run
    lcp = np.where(hi, alpha + rng.integers(0, 48, n), rng.integers(0, max(alpha, 1), n)).astype(np.uint32)
    lcp[0] = 0
    isr = rng.random(n) < p_read
    da = np.where(isr, rng.integers(0, n_reads, n), n_reads + rng.integers(0, n_refs, n)).astype(np.uint32)
    ebwt = rng.choice(np.frombuffer(symbols, dtype=np.uint8), size=n, p=sym_p).astype(np.uint8)
    return lcp, da, ebwt


def case_iid_wrap(rng):
    """Few documents, many symbols: table cells wrap mod 256 many times (:183, :247)."""
    lcp, da, ebwt = iid(rng, 20000, 3, 2, 16, 0.5, 0.5, b"ACGT")
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=3, n_refs=2, alpha=16, read_len=100, beta=0.25)


def case_long_runs(rng):
    """P(lcp>=alpha)=0.99: clusters of hundreds to thousands of symbols; with 2+2 documents
    per-cluster counts pass 255, so genome counts saturate (:96-97) and read counts wrap (:123)."""
    lcp, da, ebwt = iid(rng, 30000, 2, 2, 16, 0.99, 0.5, b"ACGTN\x00", [0.3, 0.3, 0.17, 0.17, 0.03, 0.03])
    # one run longer than a 4096-symbol tile and one that reaches EOF
    lcp[5000:14000] = 20
    lcp[-700:] = 17
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=2, n_refs=2, alpha=16, read_len=100, beta=0.1)


def case_medium(rng):
    """Medium clusters (tens to a few hundred symbols) over many documents, repeated docs."""
    lcp, da, ebwt = iid(rng, 40000, 50, 12, 16, 0.93, 0.35, b"ACGTNRY\x00",
                        [0.24, 0.24, 0.24, 0.24, 0.01, 0.01, 0.01, 0.01])
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=50, n_refs=12, alpha=16, read_len=120, beta=0.2)


def case_edges(rng):
    """lcp[0]>=alpha (leading positions ignored, :196-202); run reaching EOF (:244-245);
    reads-only and genomes-only runs; lcp==alpha; bytes '$', '#', lowercase count as 'A'."""
    n_reads, n_refs, alpha = 5, 4, 16
    lcp, da, ebwt = iid(rng, 6000, n_reads, n_refs, alpha, 0.45, 0.4, b"ACGT$#acgtN\x00")
    lcp[0:7] = [30, 16, 16, 40, 16, 3, 16]       # starts inside a run; lcp == alpha inside
    da[0:7] = [0, 5, 1, 6, 2, 7, 0]
    lcp[100:110] = 16; da[99:110] = 2             # reads-only run
    lcp[200:215] = 18; da[199:215] = n_reads + 1  # genomes-only run
    lcp[-5:] = [2, 16, 16, 17, 16]                # run reaching EOF
    da[-6:] = [1, 6, 1, 7, 2, 8]
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=n_reads, n_refs=n_refs, alpha=alpha, read_len=60, beta=0.3)


def case_iupac(rng):
    """Dense IUPAC codes with repeated documents: every branch of the cross-match block
    (:146-177) including the as-written subtraction of a zeroed operand (:154, :159)."""
    lcp, da, ebwt = iid(rng, 30000, 6, 5, 16, 0.8, 0.45, b"ACGTRYSWKMBDHVN\x00")
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=6, n_refs=5, alpha=16, read_len=200, beta=0.05)


def case_synth_c2(rng):
    """Config-C2-shaped iid synthetic (SURVEY.md 8d proportions), scaled down."""
    lcp, da, ebwt = iid(rng, 60000, 300, 25, 16, 0.40, 0.10, b"ACGTN\x00RY",
                        [0.2425, 0.2425, 0.2425, 0.2425, 0.01, 0.01, 0.005, 0.005])
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=300, n_refs=25, alpha=16, read_len=20, beta=0.1)


def case_tiny(rng):
    """Degenerate sizes: N=1, N=2 and a 5-symbol single cluster, concatenated as 3 sub-cases."""
    lcp = np.array([0, 20, 20, 17, 16], dtype=np.uint32)
    da = np.array([1, 0, 2, 0, 3], dtype=np.uint32)
    ebwt = np.frombuffer(b"ACAGT", dtype=np.uint8).copy()
    return dict(lcp=lcp, da=da, ebwt=ebwt, n_reads=2, n_refs=2, alpha=16, read_len=18, beta=0.0)


CASES = {
    "toy_text": case_toy_text, "iid_wrap": case_iid_wrap, "long_runs": case_long_runs,
    "medium": case_medium, "edges": case_edges, "iupac": case_iupac,
    "synth_c2": case_synth_c2, "tiny": case_tiny,
}


def main():
    if not os.path.exists(f"{REF}/ClusterLCP"):
        sys.exit("oracle/_ref missing: run `make -C oracle` in the build container first")
    for k, (name, fn) in enumerate(CASES.items()):
        rng = np.random.default_rng(1234 + k)
        c = fn(rng)
        ref = reference_outputs(c["lcp"], c["da"], c["ebwt"], c["n_reads"], c["n_refs"],
                                c["alpha"], c["read_len"], c["beta"])
        params = np.array([c["n_reads"], c["n_refs"], c["alpha"], c["read_len"]], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), lcp=c["lcp"], da=c["da"], ebwt=c["ebwt"],
                            params=params, beta=np.float64(c["beta"]), **ref)
        print(f"{name}: N={len(c['lcp'])} clusters={len(ref['clrs'])} "
              f"maxlen={int(ref['clrs'][:, 1].max()) if len(ref['clrs']) else 0} "
              f"nonzero_e1={int((ref['sim_e1'] > 0).sum())} size={os.path.getsize(os.path.join(HERE, name + '.npz'))}")


if __name__ == "__main__":
    main()
