"""The code that only runs with MORE THAN ONE GPU (n_dev > 1 in lime_fused_multi / lime_score_choose_multi, RCCL
communicators of more than one rank in lime_comm_*), against the oracle.  The builder's box has one GPU: every test here is
skipped there and runs the first time the suite meets a multi-GPU lease.  What they stand
for in the reference: the position-range partition of ClusterLCP.cpp:150-161 and the cluster-range partition of
ClusterBWT_DA.cpp:630-670, with the shards' tables combined (sum modulo 256) instead of shared."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _n_devices():
    try:
        from lime_amd import _lib
        return int(_lib.load().lime_device_count())
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs at least two GPUs (lime_device_count() < 2)")


@needs_two
@pytest.mark.parametrize("n_dev", [2, 3, 4])
def test_fused_multi_on_several_devices_matches_the_oracle(n_dev):
    """lime_fused_multi: ranges -> devices (a host thread each, through the staging ring), one RCCL reduce-scatter inside one
    process (ncclCommInitAll), blocks copied back; with a run that crosses a range border"""
    if _n_devices() < n_dev:
        pytest.skip(f"{n_dev} devices asked, {_n_devices()} visible")
    from lime_amd import _lib
    from oracle import oracle_py as O
    lib = _lib.load()
    n, nr, ng = 2_000_003, 3000, 70
    lcp, da, eb = O.synth(31, 0, n, nr, ng, 16, 1)
    for k in range(1, n_dev):                                   # runs across every cut (cuts are multiples of the 4096-position tile)
        cut = ((n + 4095) // 4096 * k // n_dev) * 4096
        lcp[cut - 700:cut + 900] = 40
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        sim = np.full((nr, ng), 0xAB, np.uint8)
        gnc, gml = C.c_uint64(0), C.c_uint64(0)
        rc = lib.lime_fused_multi(n_dev, None, lcp.ctypes.data, da.ctypes.data, None if e is None else e.ctypes.data, n, nr, ng, 16,
                                  sim.ctypes.data, C.byref(gnc), C.byref(gml))
        assert rc == 0, lib.lime_comm_error()
        assert (gnc.value, gml.value) == (nc, ml)
        assert np.array_equal(sim, exp), "lime_fused_multi: combined table differs from the oracle"


@needs_two
def test_score_choose_multi_on_two_devices_matches_the_single_device_result():
    """lime_score_choose_multi: the cluster list cut by position over two devices, reduce-scatter by blocks of a multiple of 16
    rows, row scan per block -- the same row maxima, offsets and (idRef, sim) lists as one device"""
    import lime_amd
    from lime_amd import _lib
    from oracle import oracle_py as O
    lib = _lib.load()
    n, nr, ng = 1_200_001, 2500, 37                             # 37: blocks do not start on a 16-byte border unless rows come in sixteens
    lcp, da, eb = O.synth(33, 0, n, nr, ng, 16, 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    norm, beta = 85, 0.02
    ctx = lime_amd.Context(0)
    for e in (eb, None):
        mx1, off1, pairs1 = ctx.score_choose(da, e, cl, nr, ng, norm, beta)
        mx = np.zeros(nr + 1, np.uint8); off = np.zeros(nr + 2, np.uint64)
        pp, npairs = C.c_void_p(), C.c_uint64(0)
        rc = lib.lime_score_choose_multi(2, None, da.ctypes.data, None if e is None else e.ctypes.data, n, cl.ctypes.data, len(cl), nr, ng,
                                         norm, C.c_float(beta), mx.ctypes.data, off.ctypes.data, C.byref(pp), C.byref(npairs))
        assert rc == 0, lib.lime_comm_error()
        pairs = ctx._pairs_out(pp, npairs)
        assert np.array_equal(mx[:nr], mx1) and np.array_equal(off[:nr + 1], off1)
        assert np.array_equal(pairs, pairs1)
    ctx.close()


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LIME_ROOT"])
import lime_amd
from lime_amd import dist as ldist
from oracle import oracle_py as O
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
comm = ldist.Comm(rank, world, dev)                              # RCCL through the C ABI, one rank per GPU
comm.check_uint8_sum_wraps()
assert comm.count() == world                                         # ncclCommCount: RCCL sees every rank
n, nr, ng = 3_000_001, 9000, 130
lcp, da, eb = O.synth(41, 0, n, nr, ng, 16, 1)
for k in range(1, world):
    cut = ((n + 4095) // 4096 * k // world) * 4096
    lcp[cut - 600:cut + 800] = 33                              # long clusters across the cuts: their updates travel as 8-byte records
lo, hi, hh = ldist.shard_ranges(n, world)[rank]
ctx = lime_amd.Context(local)
st = torch.cuda.current_stream().cuda_stream
sim_bytes = lime_amd.sim_bytes(nr, ng); blk = ldist.table_block_bytes(sim_bytes, world)
n_bins, bin_shift = ctx.records_layout(nr, ng)
per = (n_bins + world - 1) // world
if rank == 0:
    cl, enc, eml = O.detect(lcp, da, nr, 16)
for e in (eb, None):
    tl = torch.from_numpy(lcp[lo:hh].view(np.int32)).to(dev); td = torch.from_numpy(da[lo:hh].view(np.int32)).to(dev)
    te = None if e is None else torch.from_numpy(e[lo:hh]).to(dev)
    # (1) dense: private tables, one uint8 reduce-scatter
    sim = torch.zeros(blk * world, dtype=torch.uint8, device=dev); mine = torch.empty(blk, dtype=torch.uint8, device=dev)
    ctx.fused_dev(tl, td, te, hi - lo, hh - lo, hh == n, nr, ng, 16, sim, True, st)
    s, rc = ctx.stats(st); assert rc == 0, rc
    comm.reduce_scatter_tables(sim, mine, blk, st)
    nc, ml = comm.combine_counters(int(s.n_clusters), int(s.max_len))
    torch.cuda.synchronize()
    dense = [None] * world
    dist.all_gather_object(dense, mine.cpu().numpy())
    # (2) sparse: update records to the owners of their bins, twice (the second exchange reuses the grown buffers)
    for rep in range(2):
        ctx.fused_records_dev(tl, td, te, hi - lo, hh - lo, hh == n, nr, ng, 16, st)
        s2, rc = ctx.stats(st); assert rc == 0, rc
        own = torch.full((per << bin_shift,), 0xCD, dtype=torch.uint8, device=dev)
        cell_lo, nbytes = comm.exchange_records(ctx, nr, ng, own, st)
        torch.cuda.synchronize()
        parts = [None] * world
        dist.all_gather_object(parts, (cell_lo, own[:nbytes].cpu().numpy()))
        if rank == 0:
            exp = O.score(da, e, cl, nr, ng, threads=8)
            got_d = np.concatenate(dense)[:nr * ng].reshape(nr, ng)
            got_s = np.concatenate([p for _, p in sorted(parts, key=lambda x: x[0])])[:nr * ng].reshape(nr, ng)
            assert (nc, ml) == (enc, eml), ((nc, ml), (enc, eml))
            assert np.array_equal(got_d, exp), "reduce-scatter of the ranks' tables differs from the oracle"
            assert np.array_equal(got_s, exp), "the owners' blocks (exchange of update records) differ from the oracle"
ctx.close(); comm.close(); dist.destroy_process_group()
if rank == 0: print("MULTI_OK")
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@needs_two
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_ranks_reduce_scatter_and_exchange_records_against_the_oracle(world, tmp_path):
    """one process per GPU: lime_comm_reduce_scatter_tables (ncclReduceScatter, uint8 sum) and lime_comm_exchange_records
    (all-gather of the bin bases with the status words, send / receive of the slices, all-gather of the long clusters'
    records) with more than one rank"""
    if _n_devices() < world:
        pytest.skip(f"{world} devices asked, {_n_devices()} visible")
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, LIME_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(w)]
    r = subprocess.run(cmd, capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and b"MULTI_OK" in r.stdout, r.stderr.decode()[-3000:]


def test_the_multi_gpu_tests_are_listed():
    """(runs everywhere) the skip-unless-two-devices tests above exist and say why they are skipped"""
    assert needs_two.args[0] == (_n_devices() < 2)
    assert "two GPUs" in needs_two.kwargs["reason"]
