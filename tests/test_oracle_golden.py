"""The oracle (oracle/lime_oracle.c) against the reference's own outputs (tests/golden/*.npz,
made by tests/golden/make_golden.py from oracle/_ref) -- this is what PINS the oracle."""
import os
import struct

import numpy as np
import pytest

from oracle import oracle_py as O


def test_cases_present():
    from tests.conftest import GOLDEN_CASES
    assert {"toy_text", "iid_wrap", "long_runs", "medium", "edges", "iupac", "synth_c2", "tiny"} <= set(GOLDEN_CASES)


def test_detect_matches_reference(golden):
    cl, nc, ml = O.detect(golden["lcp"], golden["da"], golden["n_reads"], golden["alpha"])
    assert np.array_equal(cl, golden["clrs"])
    out = O.out_bytes(golden["n_reads"], golden["n_refs"], golden["alpha"], ml, nc)
    assert out == golden["out"].tobytes()


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_score_matches_reference(golden, ebwt_mode):
    eb = golden["ebwt"] if ebwt_mode else None
    sim = O.score(golden["da"], eb, golden["clrs"], golden["n_reads"], golden["n_refs"])
    assert np.array_equal(sim, golden[f"sim_e{ebwt_mode}"])
    sim4 = O.score(golden["da"], eb, golden["clrs"], golden["n_reads"], golden["n_refs"], threads=4)
    assert np.array_equal(sim4, sim)


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_choose_writers_match_reference(golden, ebwt_mode, tmp_path):
    sim = golden[f"sim_e{ebwt_mode}"]
    norm = (golden["read_len"] + 1 - golden["alpha"]) & 0xFFFFFFFF
    beta = np.float32(golden["beta"])
    t = str(tmp_path / "r.txt")
    O.write_res_txt(t, sim, norm, beta)
    assert open(t, "rb").read() == golden[f"txt_e{ebwt_mode}"].tobytes()
    b, p = str(tmp_path / "r.bin"), str(tmp_path / "r.pos")
    O.write_res_bin(b, p, sim, norm, beta)
    assert open(b, "rb").read() == golden[f"bin_e{ebwt_mode}"].tobytes()
    assert open(p, "rb").read() == golden[f"pos_e{ebwt_mode}"].tobytes()


def test_sym_index_table():
    exp = {ord(c): i for i, c in enumerate("ACGTRYSWKMBDHVN")}
    exp[0] = 15
    for b in range(256):
        assert O.sym_index(b) == exp.get(b, 0)


def test_synth_is_pure_function_of_index():
    a = O.synth(42, 0, 5000, 100, 7)
    b = O.synth(42, 1234, 1000, 100, 7)
    for x, y in zip(a, b):
        assert np.array_equal(x[1234:2234], y)
    assert a[0][0] == 0
    frac = (a[0] >= 16).mean()
    assert 0.35 < frac < 0.45
    assert 0.07 < (a[1] < 100).mean() < 0.13


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(O.__file__), "_ref", "ClusterLCP")),
                    reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_vs_live_reference(tmp_path):
    """Fresh random inputs through the reference binaries, compared live with the oracle."""
    import subprocess
    ref = os.path.join(os.path.dirname(O.__file__), "_ref")
    for seed, (n, nr, ng) in enumerate([(30000, 40, 9), (12000, 4, 3)]):
        lcp, da, eb = O.synth(seed + 7, 0, n, nr, ng, mode=seed % 2)
        base = str(tmp_path / f"S{seed}.fasta")
        lcp.tofile(base + ".lcp"); da.tofile(base + ".da"); eb.tofile(base + ".ebwt")
        subprocess.run([f"{ref}/ClusterLCP", base, str(nr), str(ng), "16", "1"], check=True,
                       capture_output=True, timeout=60, cwd=tmp_path)
        cl = np.fromfile(base + ".16.clrs", dtype="<u8").reshape(-1, 2)
        ocl, nc, ml = O.detect(lcp, da, nr, 16)
        assert np.array_equal(ocl, cl)
        aux = open(str(tmp_path / f"S{seed}.out"), "rb").read()
        assert aux == struct.pack("<IIIQQ", nr, ng, 16, ml, nc)
        for tag, e in (("", eb), ("_e0", None)):
            so = subprocess.run([f"{ref}/ClusterBWT_DA{tag}_small", base, "100", "0.25", "2"], check=True,
                                capture_output=True, timeout=60, cwd=tmp_path).stdout.decode()
            body = so.split("***FINAL***\n", 1)[1].split("***********", 1)[0]
            ref_sim = np.array(body.split(), dtype=np.int64).astype(np.uint8).reshape(nr, ng)
            assert np.array_equal(O.score(da, e, cl, nr, ng), ref_sim)
