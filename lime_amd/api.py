"""Host-side mirror of the reference's interface for the hot path.

The reference exposes two programs (argv + files): `ClusterLCP fileFasta numReads numGenomes
alpha threads` (src/ClusterLCP.cpp:56-71) and `ClusterBWT_DA fileFasta readLen beta threads`
(src/ClusterBWT_DA.cpp:496-529).  `cluster_lcp` / `cluster_bwt_da` below take the same
arguments and read/write the same files; `detect` / `score` / `fused` / `choose` are the
array-level calls underneath (numpy in, numpy out), all through the C ABI in
include/lime_hip.h -- i.e. through the HIP kernels.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Cluster, LimeError, Stats, check  # noqa: F401


class _LibBuf:
    """a host buffer returned by the library (uint32 words), exposed through the array interface and freed with the last array on it"""
    def __init__(self, lib, ptr, words):
        self._lib, self._ptr = lib, ptr
        self.__array_interface__ = {"data": (ptr, False), "shape": (words,), "typestr": "<u4", "version": 3}

    def __del__(self):
        try:
            self._lib.lime_free(C.c_void_p(self._ptr))
        except Exception:        # interpreter shutdown: the library handle may be gone before the last array
            pass


class Context:
    """One lime_ctx (one device).  `device=None` keeps the current HIP device."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.lime_init(-1 if device is None else int(device), C.byref(h)))
        self.h = h

    def set_option(self, key, value=""):
        """a tuning / test knob of the ctx by name (lime_set_option; include/lime_hip.h lists them); "" = the library's own choice"""
        check(self.lib.lime_set_option(self.h, str(key).encode(), str(value).encode()))

    def close(self):
        if self.h:
            self.lib.lime_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- array level, host pointers ------------------------------------------------------
    def detect(self, lcp, da, n_reads, alpha):
        """ClusterLCP scan -> (clusters u64[nC,2] ascending pStart, n_clusters, max_len)."""
        lcp = np.ascontiguousarray(lcp, dtype=np.uint32)
        da = np.ascontiguousarray(da, dtype=np.uint32)
        if len(lcp) != len(da):
            raise ValueError("lcp and da differ in length")
        out = C.c_void_p()
        nc, ml = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.lime_detect(self.h, lcp.ctypes.data, da.ctypes.data, len(lcp), n_reads, alpha,
                                   C.byref(out), C.byref(nc), C.byref(ml)))
        if nc.value:
            arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(nc.value, 2)).copy()
            self.lib.lime_free(out)
        else:
            arr = np.zeros((0, 2), dtype=np.uint64)
        return arr, int(nc.value), int(ml.value)

    def score(self, da, ebwt, clusters, n_reads, n_refs):
        """clusterAnalyze -> sim u8[n_reads, n_refs]; ebwt=None is the EBWT=0 build."""
        da = np.ascontiguousarray(da, dtype=np.uint32)
        cl = np.ascontiguousarray(clusters, dtype=np.uint64).reshape(-1, 2)
        eb = None if ebwt is None else np.ascontiguousarray(ebwt, dtype=np.uint8)
        sim = np.zeros((n_reads, n_refs), dtype=np.uint8)
        check(self.lib.lime_score(self.h, da.ctypes.data, None if eb is None else eb.ctypes.data, len(da),
                                  cl.ctypes.data if len(cl) else None, len(cl), n_reads, n_refs, sim.ctypes.data))
        return sim

    def fused(self, lcp, da, ebwt, n_reads, n_refs, alpha):
        """detect + score in one pass -> (sim, n_clusters, max_len)."""
        lcp = np.ascontiguousarray(lcp, dtype=np.uint32)
        da = np.ascontiguousarray(da, dtype=np.uint32)
        eb = None if ebwt is None else np.ascontiguousarray(ebwt, dtype=np.uint8)
        sim = np.zeros((n_reads, n_refs), dtype=np.uint8)
        nc, ml = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.lime_fused(self.h, lcp.ctypes.data, da.ctypes.data, None if eb is None else eb.ctypes.data,
                                  len(lcp), n_reads, n_refs, alpha, sim.ctypes.data, C.byref(nc), C.byref(ml)))
        return sim, int(nc.value), int(ml.value)

    def fused_stream(self, lcp, da, ebwt, n_reads, n_refs, alpha, chunk=0):
        """lime_fused through HBM in position-range chunks (copy of chunk k+1 under the scan of chunk k)."""
        lcp = np.ascontiguousarray(lcp, dtype=np.uint32)
        da = np.ascontiguousarray(da, dtype=np.uint32)
        eb = None if ebwt is None else np.ascontiguousarray(ebwt, dtype=np.uint8)
        sim = np.zeros((n_reads, n_refs), dtype=np.uint8)
        nc, ml = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.lime_fused_stream(self.h, lcp.ctypes.data, da.ctypes.data, None if eb is None else eb.ctypes.data,
                                         len(lcp), n_reads, n_refs, alpha, chunk, sim.ctypes.data, C.byref(nc), C.byref(ml)))
        return sim, int(nc.value), int(ml.value)

    def choose(self, sim):
        sim = np.ascontiguousarray(sim, dtype=np.uint8)
        nr, ng = sim.shape
        mx = np.zeros(nr, dtype=np.uint8)
        nz = np.zeros(nr, dtype=np.uint32)
        check(self.lib.lime_choose(self.h, sim.ctypes.data, nr, ng, mx.ctypes.data, nz.ctypes.data))
        return mx, nz

    def _pairs_out(self, pp, npairs):
        """the library's pair list as an array WITHOUT a copy (0.7 GB when most of configs[2]'s rows pass: zeroing + copying it cost more than the
        device side of the call); the buffer goes back to lime_free when the last view of it dies"""
        n = int(npairs.value)
        if not n or not pp.value:
            if pp.value:
                self.lib.lime_free(pp)
            return np.zeros((0, 2), dtype=np.uint32)
        return np.asarray(_LibBuf(self.lib, pp.value, 2 * n)).reshape(n, 2)

    def score_choose(self, da, ebwt, clusters, n_reads, n_refs, norm, beta, want_sim=False):
        """clusterAnalyze + clusterChoose with the table kept in HBM -> (row_max u8[n_reads],
        row_off u64[n_reads+1], pairs u32[n,2] = (idRef, sim)[, sim])."""
        da = np.ascontiguousarray(da, dtype=np.uint32)
        cl = np.ascontiguousarray(clusters, dtype=np.uint64).reshape(-1, 2)
        eb = None if ebwt is None else np.ascontiguousarray(ebwt, dtype=np.uint8)
        mx = np.zeros(n_reads + 1, dtype=np.uint8)
        off = np.zeros(n_reads + 2, dtype=np.uint64)
        sim = np.zeros((n_reads, n_refs), dtype=np.uint8) if want_sim else None
        pp, npairs = C.c_void_p(), C.c_uint64(0)
        check(self.lib.lime_score_choose(self.h, da.ctypes.data, None if eb is None else eb.ctypes.data, len(da),
                                         cl.ctypes.data if len(cl) else None, len(cl), n_reads, n_refs, norm, beta,
                                         mx.ctypes.data, off.ctypes.data, C.byref(pp), C.byref(npairs),
                                         None if sim is None else sim.ctypes.data))
        pairs = self._pairs_out(pp, npairs)
        res = (mx[:n_reads], off[:n_reads + 1], pairs)
        return res + (sim,) if want_sim else res

    def choose_pairs_dev(self, sim_t, n_reads, n_refs, norm, beta, stream=None):
        """clusterChoose of a device-resident table -> (row_max, row_off, pairs) on the host."""
        mx = np.zeros(n_reads + 1, dtype=np.uint8)
        off = np.zeros(n_reads + 2, dtype=np.uint64)
        pp, npairs = C.c_void_p(), C.c_uint64(0)
        check(self.lib.lime_choose_pairs_dev(self.h, _ptr(sim_t), n_reads, n_refs, norm, beta, mx.ctypes.data,
                                             off.ctypes.data, C.byref(pp), C.byref(npairs), stream))
        return mx[:n_reads], off[:n_reads + 1], self._pairs_out(pp, npairs)

    def fused_choose_dev(self, lcp_t, da_t, ebwt_t, n, n_reads, n_refs, alpha, norm, beta, stream=None, out=None):
        """scan + clusterAnalyze + clusterChoose on device-resident arrays, without the table where the binned path serves the pass
        -> (row_max, row_off, pairs, Stats)"""
        if out is None:                                       # out: the caller's (uint8[n_reads + 1], uint64[n_reads + 2]) to fill instead of fresh arrays
            out = (np.zeros(n_reads + 1, dtype=np.uint8), np.zeros(n_reads + 2, dtype=np.uint64))       # (9 MB of fresh pages at 10^6 reads: 0.5 ms)
        mx, off = out
        assert mx.dtype == np.uint8 and off.dtype == np.uint64 and len(mx) >= n_reads + 1 and len(off) >= n_reads + 2
        pp, npairs, s = C.c_void_p(), C.c_uint64(0), Stats()
        check(self.lib.lime_fused_choose_dev(self.h, _ptr(lcp_t), _ptr(da_t), _ptr(ebwt_t), n, n_reads, n_refs, alpha, norm, beta,
                                             mx.ctypes.data, off.ctypes.data, C.byref(pp), C.byref(npairs), C.byref(s), stream))
        return mx[:n_reads], off[:n_reads + 1], self._pairs_out(pp, npairs), s

    # ---- array level, device pointers (torch tensors on this ctx's device) ---------------
    def stats(self, stream=None):
        s = Stats()
        rc = self.lib.lime_get_stats(self.h, C.byref(s), stream)
        return s, rc

    def synth_dev(self, seed, i0, count, n_reads, n_refs, alpha, mode, lcp_t, da_t, ebwt_t, stream=None):
        check(self.lib.lime_synth_dev(self.h, seed, i0, count, n_reads, n_refs, alpha, mode,
                                      _ptr(lcp_t), _ptr(da_t), _ptr(ebwt_t), stream))

    def fused_dev(self, lcp_t, da_t, ebwt_t, n_own, n_avail, eof, n_reads, n_refs, alpha, sim_t,
                  zero_sim=True, stream=None):
        check(self.lib.lime_fused_dev(self.h, _ptr(lcp_t), _ptr(da_t), _ptr(ebwt_t), n_own, n_avail, int(eof),
                                      n_reads, n_refs, alpha, _ptr(sim_t), int(zero_sim), stream))

    # ---- owner-partitioned exchange of table updates (include/lime_hip.h) ----
    def records_layout(self, n_reads, n_refs):
        nb, sh = C.c_uint32(0), C.c_uint32(0)
        check(self.lib.lime_records_layout(self.h, n_reads, n_refs, C.byref(nb), C.byref(sh)))
        return int(nb.value), int(sh.value)

    def fused_records_dev(self, lcp_t, da_t, ebwt_t, n_own, n_avail, eof, n_reads, n_refs, alpha, stream=None):
        self._rec_shape = self.records_layout(n_reads, n_refs)
        check(self.lib.lime_fused_records_dev(self.h, _ptr(lcp_t), _ptr(da_t), _ptr(ebwt_t), n_own, n_avail, int(eof),
                                              n_reads, n_refs, alpha, stream))

    def records_get(self, stream=None):
        """-> (Records with device pointers valid until the next pass, numpy bin bases [n_bins + 1])"""
        import numpy as np
        from ._lib import Records
        nb, _ = self._rec_shape
        r = Records()
        base = np.zeros(nb + 1, dtype=np.uint64)
        check(self.lib.lime_records_get(self.h, C.byref(r), base.ctypes.data, stream))
        return r, base

    def apply_records_dev(self, n_src, rx_t, srcoff, nb, bin_shift, bigrecs_t, n_big, cell_lo, block_bytes, block_t, stream=None):
        import numpy as np
        so = np.ascontiguousarray(srcoff, dtype=np.uint64)
        check(self.lib.lime_apply_records_dev(self.h, n_src, _ptr(rx_t), so.ctypes.data, nb, bin_shift, _ptr(bigrecs_t), n_big,
                                              cell_lo, block_bytes, _ptr(block_t), stream))

    def detect_dev(self, lcp_t, da_t, n_own, n_avail, eof, pos_base, n_reads, alpha, stream=None):
        """-> (device pointer of the library-owned record list, n_clusters, max_len); the list stays valid
        until the next detect_dev on this context."""
        dc, nc, ml = C.c_void_p(), C.c_uint64(0), C.c_uint64(0)
        check(self.lib.lime_detect_dev(self.h, _ptr(lcp_t), _ptr(da_t), n_own, n_avail, int(eof), pos_base, n_reads,
                                       alpha, C.byref(dc), C.byref(nc), C.byref(ml), stream))
        return dc.value, int(nc.value), int(ml.value)

    def score_dev(self, da_t, ebwt_t, n, clusters_ptr, n_clusters, n_reads, n_refs, sim_t, zero_sim=True, stream=None):
        """clusters_ptr: device address of n_clusters lime_cluster_t records (e.g. from detect_dev)"""
        check(self.lib.lime_score_dev(self.h, _ptr(da_t), _ptr(ebwt_t), n, C.c_void_p(clusters_ptr), n_clusters, n_reads,
                                      n_refs, _ptr(sim_t), int(zero_sim), stream))

    def choose_dev(self, sim_t, n_reads, n_refs, max_t, nnz_t, stream=None):
        check(self.lib.lime_choose_dev(self.h, _ptr(sim_t), n_reads, n_refs, _ptr(max_t), _ptr(nnz_t), stream))

    def set_timing(self, on):
        check(self.lib.lime_set_timing(self.h, int(on)))

    def get_timing(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self.lib.lime_get_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    def get_timing_ex(self):
        """-> ({scan, pass, after_scan, before_scan} average ms per lime_fused_dev pass, passes timed)"""
        ms, n = (C.c_double * 4)(), C.c_uint64(0)
        check(self.lib.lime_get_timing_ex(self.h, ms, C.byref(n)))
        return {"scan": ms[0], "pass": ms[1], "after_scan": ms[2], "before_scan": ms[3]}, int(n.value)


    def host_times(self):
        """what the passes on this context cost on the host so far (lime_get_host_times)"""
        v = (C.c_double * 8)()
        check(self.lib.lime_get_host_times(self.h, v))
        return {"alloc_ms": v[0], "probe_ms": v[1], "probes": int(v[2]), "repeats": int(v[3]), "cas_fallbacks": int(v[4]),
                "records_per_symbol": None if v[5] < 0 else v[5], "choose_without_table": int(v[6])}


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def trim_cache():
    """give the device blocks that closed contexts left with the library back to the driver (lime_trim_cache); returns the bytes released"""
    return int(_lib.load().lime_trim_cache())


def reserve(nbytes):
    """take `nbytes` of device memory from the driver now, once, for the large buffers of every later context (lime_reserve); LimeError if refused"""
    rc = _lib.load().lime_reserve(int(nbytes))
    if rc != 0:
        raise LimeError(rc, "lime_reserve")


def sim_bytes(n_reads, n_refs):
    return int(_lib.load().lime_sim_bytes(n_reads, n_refs))


def aux_name(file_fasta):
    k = file_fasta.find(".fasta")               # ClusterLCP.cpp:294
    return (file_fasta if k < 0 else file_fasta[:k]) + ".out"


# ---- program level: same arguments and files as the reference's executables ----------------
def cluster_lcp(file_fasta, num_reads, num_genomes, alpha, threads=1, ctx=None):
    """`ClusterLCP fileFasta numReads numGenomes alpha threads` (src/ClusterLCP.cpp:47-320)."""
    lib = _lib.load()
    for ext in (".lcp", ".da"):
        if not os.path.exists(file_fasta + ext):
            raise FileNotFoundError(f"Error opening {file_fasta + ext}.")
    lcp = np.memmap(file_fasta + ".lcp", dtype="<u4", mode="r") if os.path.getsize(file_fasta + ".lcp") else np.zeros(0, np.uint32)
    da = np.memmap(file_fasta + ".da", dtype="<u4", mode="r") if os.path.getsize(file_fasta + ".da") else np.zeros(0, np.uint32)
    own = ctx is None
    ctx = ctx or Context()
    try:
        cl, nc, ml = ctx.detect(lcp, da[:len(lcp)], num_reads, alpha)
    finally:
        if own:
            ctx.close()
    cl = np.ascontiguousarray(cl)
    check(lib.lime_write_clrs(f"{file_fasta}.{alpha}.clrs".encode(), cl.ctypes.data if nc else None, nc))
    check(lib.lime_write_aux(aux_name(file_fasta).encode(), num_reads, num_genomes, alpha, ml, nc))
    return nc, ml


def cluster_bwt_da(file_fasta, read_len, beta, threads=1, ebwt=True, binary=True, ctx=None):
    """`ClusterBWT_DA fileFasta readLen beta threads` (src/ClusterBWT_DA.cpp:453-773).
    ebwt / binary stand for the reference's compile-time EBWT / BIN switches (Makefile:12-13)."""
    lib = _lib.load()
    nr, ng, al = C.c_uint32(), C.c_uint32(), C.c_uint32()
    ml, nc = C.c_uint64(), C.c_uint64()
    if lib.lime_read_aux(aux_name(file_fasta).encode(), C.byref(nr), C.byref(ng), C.byref(al), C.byref(ml), C.byref(nc)):
        raise FileNotFoundError(f"Error opening {aux_name(file_fasta)}.")
    if ml.value > _lib.MAX_CLUSTER:
        raise LimeError(_lib.ERR_MAXLEN, f"maximum cluster size is {ml.value} greater than sizeMaxBuf")
    read_len = int(read_len) & 0xFF                       # %hhu, :519-521
    norm = (read_len + 1 - al.value) & 0xFFFFFFFF         # :555
    clrs = np.fromfile(f"{file_fasta}.{al.value}.clrs", dtype="<u8").reshape(-1, 2)[:nc.value]
    da = np.fromfile(file_fasta + ".da", dtype="<u4")
    eb = np.fromfile(file_fasta + ".ebwt", dtype=np.uint8) if ebwt else None
    own = ctx is None
    ctx = ctx or Context()
    try:
        sim = ctx.score(da, eb, clrs, nr.value, ng.value)
        mx, _ = ctx.choose(sim) if nr.value else (np.zeros(0, np.uint8), None)
    finally:
        if own:
            ctx.close()
    beta32 = float(np.float32(beta))
    res = file_fasta + ".res"
    if binary:
        check(lib.lime_write_res_bin((res + ".bin").encode(), (res + ".pos").encode(), sim.ctypes.data,
                                     mx.ctypes.data, nr.value, ng.value, norm, beta32))
    else:
        check(lib.lime_write_res_txt((res + ".txt").encode(), sim.ctypes.data, mx.ctypes.data,
                                     nr.value, ng.value, norm, beta32))
    return sim
