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
