"""EGSAtoBCR drop-in: fastaFile.K.gesa -> .ebwt/.lcp/.da.  The expected arrays follow from
the record layout (13 bytes: u32 text, u32 suff, u32 lcp, u8 bwt; src/EGSAtoBCR.cpp:9-15,72-91); when the
reference's own build is present (oracle/_ref/EGSAtoBCR, this container) its output is compared too."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lime_amd", "bin", "EGSAtoBCR")
REF = os.path.join(ROOT, "oracle", "_ref", "EGSAtoBCR")


@pytest.mark.parametrize("n,tail", [(0, 0), (1, 0), (1000, 0), (1000, 7), (65536, 12), (200003, 1)])
def test_egsa_to_bcr(tmp_path, n, tail):
    rng = np.random.default_rng(n + tail)
    rec = np.zeros(n, dtype=np.dtype([("text", "<u4"), ("suff", "<u4"), ("lcp", "<u4"), ("bwt", "u1")]))
    rec["text"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    rec["suff"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    rec["lcp"] = rng.integers(0, 300, n).astype(np.uint32)
    rec["bwt"] = rng.choice(np.frombuffer(b"ACGTN$\x00", dtype=np.uint8), n)
    assert rec.dtype.itemsize == 13
    raw = rec.tobytes() + bytes(rng.integers(0, 256, tail, dtype=np.uint8))

    def run(exe, d):
        os.makedirs(d, exist_ok=True)
        base = os.path.join(d, "x.fasta")
        open(base + ".5.gesa", "wb").write(raw)
        p = subprocess.run([exe, base, "5"], capture_output=True, timeout=60)
        assert p.returncode == 0, p.stderr
        return [open(base + e, "rb").read() for e in (".ebwt", ".lcp", ".da")], p.stderr

    (eb, lc, da), err = run(EXE, str(tmp_path / "ours"))
    assert eb == rec["bwt"].tobytes() and lc == rec["lcp"].tobytes() and da == rec["text"].tobytes()
    assert f"The total number of elements is {n}".encode() in err
    if os.path.exists(REF):
        ref_out, _ = run(REF, str(tmp_path / "ref"))
        assert [eb, lc, da] == ref_out


def test_egsa_to_bcr_usage(tmp_path):
    assert subprocess.run([EXE], capture_output=True).returncode == 1
    assert subprocess.run([EXE, str(tmp_path / "missing.fasta"), "3"], capture_output=True).returncode != 0


def test_suffix_sort_builder_matches_the_naive_one():
    """lime_amd.builder.build_arrays_sa (prefix doubling + Kasai) == the naive sort on a toy collection with repeats,
    a region shared by two genomes and IUPAC codes"""
    import numpy as np
    from lime_amd.builder import build_arrays, build_arrays_sa
    rng = np.random.default_rng(7)

    def rs(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
    g = [rs(500), rs(350) + b"NNRYK" + rs(60)]
    g[1] = g[1][:120] + g[0][40:160] + g[1][240:]
    reads = []
    for _ in range(60):
        src = g[int(rng.integers(0, 2))]
        s = int(rng.integers(0, len(src) - 30))
        reads.append(src[s:s + 30])
    reads += [reads[0], reads[1]]                        # identical reads
    for term in (0, ord("$")):
        a = build_arrays(reads, g, term=term)
        b = build_arrays_sa(reads, g, term=term)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_truncated_lcp_gives_the_same_clusters():
    """README.md:59-61: an lcp array truncated at k >= alpha (eGap --trlcp k) is as good as the full one for LiME --
    detection only asks lcp >= alpha"""
    import numpy as np
    from tests.conftest import load_golden
    from oracle import oracle_py as O
    g = load_golden("text_example")
    full = O.detect(g["lcp"], g["da"], g["n_reads"], g["alpha"])
    for k in (g["alpha"], g["alpha"] + 1, 40):
        cut = O.detect(np.minimum(g["lcp"], k).astype(np.uint32), g["da"], g["n_reads"], g["alpha"])
        assert np.array_equal(cut[0], full[0]) and cut[1:] == full[1:]
    below = O.detect(np.minimum(g["lcp"], g["alpha"] - 1).astype(np.uint32), g["da"], g["n_reads"], g["alpha"])
    assert below[1] == 0                                  # truncated below alpha: nothing left, as expected
