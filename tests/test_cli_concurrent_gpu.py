"""LiME_paired.sh starts four ClusterLCP processes at once (LiME_paired.sh:44-53), one per dataset (1F, 1RC, 2F, 2RC):
four concurrent drop-in processes on this box's GPU (four HIP contexts, staging rings, device buffers side by side),
every output byte-compared with the oracle's records."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lime_amd", "bin")


def test_four_concurrent_clusterlcp_processes(tmp_path):
    nr, ng, alpha = 5000, 40, 16
    cases = []
    for k, n in enumerate((3_000_001, 2_500_000, 3_300_000, 1_234_567)):
        lcp, da, eb = O.synth(100 + k, 0, n, nr, ng, alpha, k & 1)
        base = str(tmp_path / f"D{k}.fasta")
        lcp.tofile(base + ".lcp"); da.tofile(base + ".da")
        cases.append((base, lcp, da, n))
    env = dict(os.environ, LIME_DETECT_CHUNK="1048576", LIME_FORCE_STAGING="1")   # several chunks per process through the staging ring
    procs = [subprocess.Popen([f"{BIN}/ClusterLCP", base, str(nr), str(ng), str(alpha), "4"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, env=env, cwd=tmp_path) for base, _, _, _ in cases]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    for k, (base, lcp, da, n) in enumerate(cases):
        cl, nc, ml = O.detect(lcp, da, nr, alpha)
        assert open(f"{base}.{alpha}.clrs", "rb").read() == cl.astype("<u8").tobytes()
        assert open(str(tmp_path / f"D{k}.out"), "rb").read() == O.out_bytes(nr, ng, alpha, ml, nc)


def test_pick_device_is_stable_and_in_range():
    from lime_amd import _lib
    lib = _lib.load()
    n = lib.lime_device_count()
    for salt in (0, 1, 7, 12345):
        assert 0 <= lib.lime_pick_device(salt) < max(n, 1)


@pytest.mark.parametrize("io_threads", [3, 7])
def test_staging_ring_with_odd_sizes_and_thread_counts(monkeypatch, io_threads):
    """chunks of 4 Mi symbols (pieces of 16 MB and more, copied into the pinned slots by several threads) whose byte
    counts are not multiples of the thread count: every byte must arrive (a rounding slip once left the last few
    bytes of a piece uncopied)"""
    import lime_amd
    monkeypatch.setenv("LIME_IO_THREADS", str(io_threads))
    n, nr, ng = 9_000_003, 3000, 50
    lcp, da, eb = O.synth(31, 0, n, nr, ng, 16, 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, eb, cl, nr, ng, threads=8)
    c = lime_amd.Context()
    try:
        gcl, gnc, gml = c.detect(lcp, da, nr, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(gcl, cl)
        assert np.array_equal(c.score(da, eb, cl, nr, ng), exp)
        sim, gnc, gml = c.fused_stream(lcp, da, eb, nr, ng, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    finally:
        c.close()
