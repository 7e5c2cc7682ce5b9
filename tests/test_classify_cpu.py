"""Read assignment: lime_classify / the drop-in Classify program against the bytes the
reference's own Classify builds wrote (tests/golden/classify_*.npz, made by make_golden_classify.py),
for BIN=1/0, HIGHER=0/1, single- and paired-end inputs and four taxonomic ranks.  Host code only."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "classify_*.npz")))
RANKS = (0, 1, 2, 4)


@pytest.fixture(scope="module")
def lib():
    from lime_amd import _lib
    return _lib.load()


def _write_inputs(lib, g, td):
    sims, norm, beta = g["sims"], int(g["norm"]), float(g["beta"])
    bases = []
    for i, s in enumerate(sims):
        s = np.ascontiguousarray(s)
        b = os.path.join(td, f"in{i}.res")
        assert lib.lime_write_res_txt((b + ".txt").encode(), s.ctypes.data, None, s.shape[0], s.shape[1], norm, C.c_float(beta)) == 0
        assert lib.lime_write_res_bin((b + ".bin").encode(), (b + ".pos").encode(), s.ctypes.data, None, s.shape[0],
                                      s.shape[1], norm, C.c_float(beta)) == 0
        bases.append(b)
    tax = os.path.join(td, "lineage.csv")
    open(tax, "wb").write(g["tax"].tobytes())
    return bases, tax


@pytest.mark.parametrize("case", CASES, ids=[os.path.basename(c)[9:-4] for c in CASES])
def test_classify_matches_reference(lib, case, tmp_path):
    g = np.load(case)
    bases, tax = _write_inputs(lib, g, str(tmp_path))
    n_files, n_reads, n_targ = g["sims"].shape
    arr = (C.c_char_p * n_files)(*[b.encode() for b in bases])
    for binary in (1, 0):
        for higher in (0, 1):
            for rank in RANKS:
                key = f"out_b{binary}_h{higher}_r{rank}"
                if key not in g.files:
                    continue
                outp = str(tmp_path / f"o_{binary}{higher}{rank}.txt")
                counts = (C.c_uint64 * 4)()
                rc = lib.lime_classify(n_files, arr, binary, n_reads, n_targ, outp.encode(), tax.encode(), rank, higher, counts)
                assert rc == 0, lib.lime_classify_error()
                got = open(outp, "rb").read()
                assert got == g[key].tobytes(), key
                lines = got.decode().splitlines()[1:]
                assert [sum(1 for l in lines if l[0] == t) for t in "CUAH"] == list(counts)


def test_classify_program_is_a_dropin(lib, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "classify_paired.npz"))
    bases, tax = _write_inputs(lib, g, str(tmp_path))
    n_files, n_reads, n_targ = g["sims"].shape
    exe = os.path.join(ROOT, "lime_amd", "bin", "Classify")
    for binary, higher, rank in ((1, 0, 1), (0, 0, 2), (1, 1, 1), (0, 1, 4)):
        outp = str(tmp_path / f"p_{binary}{higher}{rank}.txt")
        env = dict(os.environ, LIME_BIN=str(binary), LIME_HIGHER=str(higher))
        p = subprocess.run([exe, str(n_files)] + bases + [str(n_reads), str(n_targ), outp, tax, str(rank), "4"],
                           capture_output=True, env=env, timeout=60)
        assert p.returncode == 0, p.stderr
        assert open(outp, "rb").read() == g[f"out_b{binary}_h{higher}_r{rank}"].tobytes()
        assert b"Number of successfully classified reads" in p.stdout
    # usage errors exit 1 like the reference (Classify.cpp:321-339)
    assert subprocess.run([exe], capture_output=True).returncode == 1
    assert subprocess.run([exe, "3", "a", "b", "c", "1", "1", "o", "t", "1", "1"], capture_output=True).returncode == 1


def test_classify_rejects_short_taxonomy(lib, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "classify_single.npz"))
    bases, tax = _write_inputs(lib, g, str(tmp_path))
    n_files, n_reads, n_targ = g["sims"].shape
    arr = (C.c_char_p * n_files)(*[b.encode() for b in bases])
    rc = lib.lime_classify(n_files, arr, 1, n_reads, n_targ + 1, str(tmp_path / "o").encode(), tax.encode(), 1, 0, None)
    assert rc != 0 and b"poor taxonomy" in lib.lime_classify_error()
