"""ctypes binding of oracle/liblime_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (lime_amd) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Cluster(C.Structure):
    _fields_ = [("pStart", C.c_uint64), ("len", C.c_uint64)]


def build():
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _DIR, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_DIR, "liblime_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u32p, u8p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
        L.lime_oracle_detect.argtypes = [u32p, u32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.POINTER(_Cluster)), u64p, u64p]
        L.lime_oracle_detect.restype = C.c_int
        L.lime_oracle_score.argtypes = [u32p, u8p, C.c_uint64, C.c_void_p, C.c_uint64,
                                        C.c_uint32, C.c_uint32, u8p, C.c_int]
        L.lime_oracle_score.restype = C.c_int
        L.lime_oracle_pair_score.argtypes = [u8p, u8p]
        L.lime_oracle_pair_score.restype = C.c_uint8
        L.lime_oracle_sym_index.argtypes = [C.c_uint8]
        L.lime_oracle_sym_index.restype = C.c_uint8
        L.lime_oracle_choose.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p, u32p]
        L.lime_oracle_choose.restype = None
        L.lime_oracle_write_res_txt.argtypes = [C.c_char_p, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]
        L.lime_oracle_write_res_txt.restype = C.c_int
        L.lime_oracle_write_res_bin.argtypes = [C.c_char_p, C.c_char_p, u8p, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_float]
        L.lime_oracle_write_res_bin.restype = C.c_int
        L.lime_oracle_synth.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint32, u32p, u32p, u8p]
        L.lime_oracle_synth.restype = None
        L.lime_oracle_free.argtypes = [C.c_void_p]
        L.lime_oracle_free.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def detect(lcp, da, n_reads, alpha):
    """-> (clusters u64[nC,2] ascending pStart, n_clusters, max_len)"""
    lcp = np.ascontiguousarray(lcp, dtype=np.uint32)
    da = np.ascontiguousarray(da, dtype=np.uint32)
    out = C.POINTER(_Cluster)()
    nc, ml = C.c_uint64(0), C.c_uint64(0)
    rc = lib().lime_oracle_detect(_p(lcp, C.c_uint32), _p(da, C.c_uint32), len(lcp), n_reads, alpha,
                                  C.byref(out), C.byref(nc), C.byref(ml))
    if rc:
        raise MemoryError("lime_oracle_detect")
    if nc.value:
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(nc.value, 2)).copy()
    else:
        arr = np.zeros((0, 2), dtype=np.uint64)
    lib().lime_oracle_free(out)
    return arr, int(nc.value), int(ml.value)


def score(da, ebwt, clusters, n_reads, n_refs, threads=1):
    """-> sim u8[n_reads, n_refs].  ebwt=None selects the EBWT=0 arithmetic."""
    da = np.ascontiguousarray(da, dtype=np.uint32)
    cl = np.ascontiguousarray(clusters, dtype=np.uint64).reshape(-1, 2)
    sim = np.zeros((n_reads, n_refs), dtype=np.uint8)
    eb = None
    if ebwt is not None:
        eb = np.ascontiguousarray(ebwt, dtype=np.uint8)
    rc = lib().lime_oracle_score(_p(da, C.c_uint32), _p(eb, C.c_uint8) if eb is not None else None,
                                 len(da), cl.ctypes.data, len(cl), n_reads, n_refs,
                                 _p(sim, C.c_uint8), threads)
    if rc:
        raise ValueError(f"lime_oracle_score rc={rc}")
    return sim


def pair_score(cr, cg):
    cr = np.ascontiguousarray(cr, dtype=np.uint8)
    cg = np.ascontiguousarray(cg, dtype=np.uint8)
    return int(lib().lime_oracle_pair_score(_p(cr, C.c_uint8), _p(cg, C.c_uint8)))


def sym_index(b):
    return int(lib().lime_oracle_sym_index(b))


def choose(sim):
    sim = np.ascontiguousarray(sim, dtype=np.uint8)
    nr, ng = sim.shape
    mx = np.zeros(nr, dtype=np.uint8)
    nz = np.zeros(nr, dtype=np.uint32)
    lib().lime_oracle_choose(_p(sim, C.c_uint8), nr, ng, _p(mx, C.c_uint8), _p(nz, C.c_uint32))
    return mx, nz


def write_res_txt(path, sim, norm, beta):
    sim = np.ascontiguousarray(sim, dtype=np.uint8)
    rc = lib().lime_oracle_write_res_txt(path.encode(), _p(sim, C.c_uint8), sim.shape[0], sim.shape[1],
                                         norm & 0xFFFFFFFF, beta)
    if rc:
        raise OSError(path)


def write_res_bin(path_bin, path_pos, sim, norm, beta):
    sim = np.ascontiguousarray(sim, dtype=np.uint8)
    rc = lib().lime_oracle_write_res_bin(path_bin.encode(), path_pos.encode(), _p(sim, C.c_uint8),
                                         sim.shape[0], sim.shape[1], norm & 0xFFFFFFFF, beta)
    if rc:
        raise OSError(path_bin)


def synth(seed, i0, count, n_reads, n_refs, alpha=16, mode=0):
    lcp = np.empty(count, dtype=np.uint32)
    da = np.empty(count, dtype=np.uint32)
    eb = np.empty(count, dtype=np.uint8)
    lib().lime_oracle_synth(seed, i0, count, n_reads, n_refs, alpha, mode,
                            _p(lcp, C.c_uint32), _p(da, C.c_uint32), _p(eb, C.c_uint8))
    return lcp, da, eb


def out_bytes(n_reads, n_refs, alpha, max_len, n_clusters):
    """The 28-byte aux .out file, src/ClusterLCP.cpp:304-308."""
    import struct
    return struct.pack("<IIIQQ", n_reads, n_refs, alpha, max_len, n_clusters)
