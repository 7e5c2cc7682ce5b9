"""Full-size bit-exactness against the reference's OWN programs.

The oracle-checked GPU tests stop at 4*10^7 symbols; the kernels that serve 10^9 .. 10^10 symbols (k_part_lines, k_part<.., true>, k_apply_tiles<true>,
two-level bins of hundreds of tiles, several sub-regions per wave) are other code paths, which the full-size tests compared only with the library
itself.  Here the reference's ClusterLCP and ClusterBWT_DA (oracle/_ref, compiled from /root/reference by oracle/Makefile; they travel to the GPU
box as binaries) run on the SAME arrays, written out as their input files, and every output file is compared byte for byte:

  <base>.out            28 bytes: numReads, numGenomes, alpha, maxLen, nClusters            ClusterLCP.cpp:294-310
  <file>.16.clrs        (pStart, len) records, 1 thread = ascending pStart                  ClusterLCP.cpp:229-235
  <file>.res.bin/.pos   clusterChoose's lists of the reads that pass beta                   ClusterBWT_DA.cpp:376-436

against lime_detect_dev's records and lime_fused_choose_dev + lime_write_res_bin_pairs, with and without the table.  Skipped where oracle/_ref is absent.
"""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
ALPHA, READ_LEN = 16, 100
NORM = READ_LEN + 1 - ALPHA                       # ClusterBWT_DA.cpp:555

CASES = {
    # BASELINE.json configs[2] as it stands: 299 bins of 256 regions, k_part_lines, k_apply_tiles<false> with 8 lanes a run (1193 bins of 64, k_part until round 6).  beta 0.02: three quarters of the rows pass (9*10^7 pairs)
    "C3": dict(n=1_000_000_000, nr=1_000_000, ng=5000, ebwt=0, mode=0, beta=0.02, records=1e8),
    # the first 2*10^9 symbols of the north_star series' collection (10^6 x 1000 = 1 GB table): 477 bins of 32 regions, k_part_lines,
    # k_apply_tiles<true> (2.4*10^8 records).  beta 0.04: a row passes with a cell >= 4
    "N1E10_SLICE": dict(n=2_000_000_000, nr=1_000_000, ng=1000, ebwt=0, mode=0, beta=0.04, records=2e8),
    # the first 10^9 symbols of configs[4]'s shape on the clustered generator, the reference's default build (EBWT=1): a 10.3 GB table = three
    # sub-regions per scan wave, 308 bins of 512 regions (1229 of 128 until round 6), 1.7*10^8 records
    "C5_CLUSTERED_SLICE": dict(n=1_000_000_000, nr=3_000_000, ng=3423, ebwt=1, mode=1, beta=0.02, records=1e8),
}


def _to_file(t, path, count):
    with open(path, "wb") as f:                   # in pieces: no second copy of the array on the host
        for lo in range(0, count, 1 << 27):
            f.write(t[lo:min(count, lo + (1 << 27))].cpu().numpy().tobytes())


def _same_file(a, b):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(1 << 26), fb.read(1 << 26)
            if x != y:
                return False
            if not x:
                return True


@pytest.mark.parametrize("case", list(CASES))
def test_reference_programs_at_full_size(case):
    import torch
    import lime_amd
    from lime_amd import _lib
    cfg = CASES[case]
    bwt = "ClusterBWT_DA" if cfg["ebwt"] else "ClusterBWT_DA_e0"
    if not (os.path.exists(f"{REF}/ClusterLCP") and os.path.exists(f"{REF}/{bwt}")):
        pytest.skip("oracle/_ref (the reference's programs, built where /root/reference exists) is not in this checkout")
    n, nr, ng, beta = cfg["n"], cfg["nr"], cfg["ng"], cfg["beta"]
    lime_amd.trim_cache()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    c = lime_amd.Context()
    try:
        lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
        eb = torch.empty(n, dtype=torch.uint8, device=dev) if cfg["ebwt"] else None
        c.synth_dev(42, 0, n, nr, ng, ALPHA, cfg["mode"], lcp, da, eb)
        torch.cuda.synchronize()
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            base = os.path.join(td, "S.fasta")
            _to_file(lcp, base + ".lcp", n); _to_file(da, base + ".da", n)
            if eb is not None:
                _to_file(eb, base + ".ebwt", n)
            # ---- the reference: ClusterLCP with ONE thread (its record order is the threads' arrival order otherwise, ClusterLCP.cpp:229-235),
            # ClusterBWT_DA with four (README.md:145; the table does not depend on the thread count)
            subprocess.run([f"{REF}/ClusterLCP", base, str(nr), str(ng), str(ALPHA), "1"], check=True, capture_output=True, cwd=td, timeout=900)
            subprocess.run([f"{REF}/{bwt}", base, str(READ_LEN), repr(beta), "4"], check=True, capture_output=True, cwd=td, timeout=1500)
            ref_out = open(os.path.join(td, "S.out"), "rb").read()
            _nr, _ng, _al, ref_ml, ref_nc = struct.unpack("<IIIQQ", ref_out)
            # ---- detection: .out and .clrs
            ptr, nc, ml = c.detect_dev(lcp, da, n, n, True, 0, nr, ALPHA)
            assert struct.pack("<IIIQQ", nr, ng, ALPHA, ml, nc) == ref_out, ((nc, ml), (ref_nc, ref_ml))
            rec = torch.empty((nc, 2), dtype=torch.int64, device=dev)
            assert _lib.hip_memcpy_d2d(rec.data_ptr(), ptr, nc * 16) == 0
            ref_clrs = np.fromfile(f"{base}.{ALPHA}.clrs", dtype="<u8")
            assert ref_clrs.size == 2 * nc
            for lo in range(0, nc, 1 << 25):      # piecewise: 3 GB of records at 2*10^9 symbols
                hi = min(nc, lo + (1 << 25))
                assert np.array_equal(rec[lo:hi].cpu().numpy().view(np.uint64).ravel(), ref_clrs[2 * lo:2 * hi]), f".clrs differs in records {lo}..{hi}"
            del rec, ref_clrs
            # ---- scoring + clusterChoose: .res.bin / .res.pos, through the table and without it
            beta32 = float(np.float32(beta))
            for free in ("1", "0"):
                c.set_option("choose_free", free)
                mx, off, pairs, s = c.fused_choose_dev(lcp, da, eb, n, nr, ng, ALPHA, NORM, beta32)
                assert (s.n_clusters, s.max_len) == (ref_nc, ref_ml) and s.flags == 0
                assert s.n_updates >= cfg["records"], int(s.n_updates)         # (the kernels this case is here for did run)
                pb = np.ascontiguousarray(pairs)
                got_bin, got_pos = os.path.join(td, f"got{free}.bin"), os.path.join(td, f"got{free}.pos")
                mxp = np.zeros(nr + 1, np.uint8); mxp[:nr] = mx
                offp = np.ascontiguousarray(off, dtype=np.uint64)
                assert lib.lime_write_res_bin_pairs(got_bin.encode(), got_pos.encode(), mxp.ctypes.data, offp.ctypes.data,
                                                    pb.ctypes.data if len(pb) else None, nr, NORM, beta32) == 0
                assert len(pairs) > 0 and int((np.diff(off.astype(np.int64)) > 0).sum()) < nr   # some rows pass, not all
                assert _same_file(got_pos, base + ".res.pos"), f".res.pos differs (choose_free {free})"
                assert _same_file(got_bin, base + ".res.bin"), f".res.bin differs (choose_free {free})"
                del mx, off, pairs, pb
            assert c.host_times()["choose_without_table"] == 1
    finally:
        c.close()
        lime_amd.trim_cache()
