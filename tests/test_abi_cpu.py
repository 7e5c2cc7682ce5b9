"""CPU-side checks of the product library: it loads, exports every symbol the header
declares, fails loudly without a GPU, and its pure host helpers (same source as the device
routines, lime_amd/csrc/lime_device.h) agree with the oracle.  No GPU compute here."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

from lime_amd import _lib
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "lime_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lime_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lime_hip.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.lime_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.lime_init(-1, C.byref(h))
    assert rc == _lib.ERR_HIP and not h
    assert b"no CPU path" in lib.lime_last_error()
    import lime_amd
    with pytest.raises(lime_amd.LimeError):
        lime_amd.Context()


def test_sym_index_matches_oracle():
    lib = _lib.load()
    for b in range(256):
        assert lib.lime_sym_index(b) == O.sym_index(b), b


def test_pair_score_matches_oracle():
    lib = _lib.load()
    rng = np.random.default_rng(5)
    n_bad = 0
    for k in range(20000):
        kind = k % 4
        if kind == 0:      # plain bases only
            cr = np.zeros(16, np.uint8); cg = np.zeros(16, np.uint8)
            cr[:4] = rng.integers(0, 6, 4); cg[:4] = rng.integers(0, 6, 4)
        elif kind == 1:    # sparse IUPAC
            cr = (rng.integers(0, 4, 16) * (rng.random(16) < 0.3)).astype(np.uint8)
            cg = (rng.integers(0, 4, 16) * (rng.random(16) < 0.3)).astype(np.uint8)
        elif kind == 2:    # dense, small counts
            cr = rng.integers(0, 5, 16).astype(np.uint8); cg = rng.integers(0, 5, 16).astype(np.uint8)
        else:              # full byte range: wrap of t
            cr = rng.integers(0, 256, 16).astype(np.uint8); cg = rng.integers(0, 256, 16).astype(np.uint8)
        a = lib.lime_pair_score(cr.ctypes.data, cg.ctypes.data)
        b = O.pair_score(cr, cg)
        n_bad += a != b
    assert n_bad == 0


def test_single_symbol_pairs_match_oracle():
    """one-hot histograms: the value the in-tile fast path takes from iupac_match()."""
    lib = _lib.load()
    for a in range(16):
        for b in range(16):
            cr = np.zeros(16, np.uint8); cg = np.zeros(16, np.uint8)
            cr[a] = 1; cg[b] = 1
            exp = O.pair_score(cr, cg)
            assert lib.lime_pair_score(cr.ctypes.data, cg.ctypes.data) == exp
            assert exp in (0, 1)


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_file_writers_match_reference_bytes(golden, ebwt_mode, tmp_path):
    lib = _lib.load()
    sim = np.ascontiguousarray(golden[f"sim_e{ebwt_mode}"])
    nr, ng = sim.shape
    norm = (golden["read_len"] + 1 - golden["alpha"]) & 0xFFFFFFFF
    beta = float(np.float32(golden["beta"]))
    t = str(tmp_path / "x.txt").encode()
    assert lib.lime_write_res_txt(t, sim.ctypes.data, None, nr, ng, norm, beta) == 0
    assert open(t, "rb").read() == golden[f"txt_e{ebwt_mode}"].tobytes()
    b, p = str(tmp_path / "x.bin").encode(), str(tmp_path / "x.pos").encode()
    mx = sim.max(axis=1).astype(np.uint8) if ng else np.zeros(nr, np.uint8)
    assert lib.lime_write_res_bin(b, p, sim.ctypes.data, mx.ctypes.data, nr, ng, norm, beta) == 0
    assert open(b, "rb").read() == golden[f"bin_e{ebwt_mode}"].tobytes()
    assert open(p, "rb").read() == golden[f"pos_e{ebwt_mode}"].tobytes()


def test_clrs_and_aux_writers(golden, tmp_path):
    lib = _lib.load()
    cl = np.ascontiguousarray(golden["clrs"])
    f = str(tmp_path / "c.clrs").encode()
    assert lib.lime_write_clrs(f, cl.ctypes.data if len(cl) else None, len(cl)) == 0
    assert open(f, "rb").read() == cl.tobytes()
    a = str(tmp_path / "c.out").encode()
    ml = int(cl[:, 1].max()) if len(cl) else 0
    assert lib.lime_write_aux(a, golden["n_reads"], golden["n_refs"], golden["alpha"], ml, len(cl)) == 0
    assert open(a, "rb").read() == golden["out"].tobytes()
    nr, ng, al, m, n = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    assert lib.lime_read_aux(a, C.byref(nr), C.byref(ng), C.byref(al), C.byref(m), C.byref(n)) == 0
    assert (nr.value, ng.value, al.value, m.value, n.value) == (golden["n_reads"], golden["n_refs"], golden["alpha"], ml, len(cl))


def test_combine_edges_host_logic():
    """lime_combine_edges (pure host): runs longer than the halo across shard borders -- closed by a later shard's
    first head or by the end of the collection; only a read+genome run is an error (ClusterBWT_DA.cpp:558-562)"""
    import ctypes as C
    import numpy as np
    from lime_amd import _lib
    lib = _lib.load()
    HEAD, LR, LG, OPEN, OR, OG = 1, 2, 4, 8, 16, 32

    def rc(*e):
        a = np.array(e, dtype=np.uint32)
        return lib.lime_combine_edges(a.ctypes.data, len(a))
    assert rc() == 0 and rc(HEAD) == 0
    assert rc(HEAD | OPEN | OG, HEAD | LG) == 0                       # genome-only run closed in the next shard
    assert rc(HEAD | OPEN | OG, HEAD | LG | LR) == -4                 # ... a read joins before it closes: a cluster
    assert rc(HEAD | OPEN | OR, LR, LR | LG | HEAD) == -4             # through a shard without any head
    assert rc(HEAD | OPEN | OR, LR, LR | HEAD) == 0
    assert rc(HEAD | OPEN | OR, LG) == -4                             # closed by the end of the collection
    assert rc(HEAD | OPEN | OR, LR) == 0
    assert rc(HEAD | LR | LG, HEAD | LR | LG) == 0                    # leading content without an open run: nothing
    assert rc(HEAD | OPEN | OG, HEAD | LG | OPEN | OR, HEAD | LR) == 0  # two separate long runs


def test_scan_resources():
    """the scan kernels' register / LDS budget (tools/kres.py --gate through `make resources`): planned waves per SIMD, no VGPR spills or
    scratch, the LDS of one workgroup fits a CU.  A performance property: gated here, not in the build."""
    import shutil
    import subprocess
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lime_amd", "csrc"), "-s", "resources"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]


def test_exec_lint_catches_a_shuffle_under_a_condition(tmp_path):
    """tools/exec_lint.py (part of `make resources`, so test_scan_resources runs it on HEAD): round 5's bug put back into a copy of the kernels --
    k_apply_tiles<true> fetching its index entries with `lx < nl ? __shfl(a, lx) : 0`, a shuffle the compiler puts under a branch, so that source
    lanes outside the condition deliver 0 (commit 0b63920 fixed it) -- must fail the lint, in exactly that kernel"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = open(os.path.join(ROOT, "lime_amd", "csrc", "lime_kernels.hip")).read()
    fixed = ("            const uint32_t sa = (uint32_t)__shfl((int)a_, (int)(lx & 63u)), se = (uint32_t)__shfl((int)e_, (int)(lx & 63u));\n"
             "            s.fax = lx < nl_ ? sa : 0u; s.fex = lx < nl_ ? se : 0u;")
    buggy = "            s.fax = lx < nl_ ? (uint32_t)__shfl((int)a_, (int)lx) : 0u; s.fex = lx < nl_ ? (uint32_t)__shfl((int)e_, (int)lx) : 0u;"
    assert src.count(fixed) == 1, "the site the test puts the bug back into has moved: update the test"
    (tmp_path / "lime_kernels.hip").write_text(src.replace(fixed, buggy))
    asm = tmp_path / "lime_kernels.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "--offload-arch=gfx950", "-Wno-unused-function", "-w",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "lime_amd", "csrc"), "--cuda-device-only", "-S",
                        "-gline-tables-only", str(tmp_path / "lime_kernels.hip"), "-o", str(asm)], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exec_lint.py"), str(asm), "--allow", os.path.join(ROOT, "tools", "exec_lint_allow.txt")],
                       capture_output=True, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 1 and "EXEC LINT FAILED" in out, out[-2000:]
    flagged = [ln for ln in out.splitlines() if ln.startswith("  ")]
    assert flagged and all("k_apply_tiles<1," in ln and "ds_bpermute_b32" in ln for ln in flagged), out[-2000:]


def test_exec_lint_follows_saved_copies_and_renamed_else_masks():
    """tools/exec_lint.py's model of EXEC on the two instruction patterns round 6 taught it (before, the first one left a restriction on the stack
    for the rest of the kernel -- 769 "sites" in the scan kernels that ran with all lanes -- and the second one made new ones): a region run by a
    precomputed lane set between `s_mov D, exec` and `s_or exec, exec, D`, and an if / else whose other half is kept in ANOTHER register than the
    saved mask.  Inside the regions a cross-lane operation is reported, after them it is not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exec_lint as L

    def restricted_at(lines):
        st, out = (), []
        for ln in lines:
            out.append(any(not x.startswith("~") for x in st))
            st = L.step(st, ln)
        return out, st
    r, st = restricted_at(["s_mov_b64 s[18:19], exec", "s_mov_b64 exec, s[8:9]", "ds_bpermute_b32 v1, v2, v3", "s_or_b64 exec, exec, s[18:19]",
                           "ds_bpermute_b32 v1, v2, v3"])
    assert r == [False, False, True, True, False] and st == ()
    r, st = restricted_at(["s_and_saveexec_b64 s[8:9], s[84:85]", "s_xor_b64 s[86:87], exec, s[8:9]", "ds_bpermute_b32 v1, v2, v3",
                           "s_andn2_saveexec_b64 s[86:87], s[86:87]", "ds_bpermute_b32 v1, v2, v3", "s_or_b64 exec, exec, s[86:87]", "ds_bpermute_b32 v1, v2, v3"])
    assert r == [False, True, True, True, True, True, False] and st == ()
    # the usual shape (one register) and a copy taken INSIDE an open region stay what they were
    r, st = restricted_at(["s_and_saveexec_b64 s[4:5], vcc", "s_mov_b64 s[6:7], exec", "s_mov_b64 exec, s[10:11]", "s_or_b64 exec, exec, s[6:7]",
                           "v_mov_b32_dpp v1, v2 row_shr:1", "s_or_b64 exec, exec, s[4:5]", "v_mov_b32_dpp v1, v2 row_shr:1"])
    assert r == [False, True, True, True, True, True, False] and st == ()


def test_library_buffers_become_arrays_without_a_copy_and_are_freed_with_the_last_view():
    """lime_amd.api._LibBuf (the (idRef, sim) lists of clusterChoose come back as the library's own buffer): the array sees the buffer's words,
    is writable, and lime_free runs exactly once, when the last view dies"""
    from lime_amd import api
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
    freed = []

    class Lib:
        def lime_free(self, p):
            freed.append(p.value); libc.free(p)
    ptr = libc.malloc(80)
    C.memmove(ptr, (C.c_uint32 * 20)(*range(20)), 80)
    a = np.asarray(api._LibBuf(Lib(), ptr, 20)).reshape(10, 2)
    assert a.dtype == np.uint32 and a.flags.writeable and a[3, 1] == 7 and a.ctypes.data == ptr
    b = a[2:4]
    del a
    assert not freed and b.sum() == 4 + 5 + 6 + 7
    del b
    assert freed == [ptr]
