"""Two ranks on ONE GPU (gloo control plane, every rank on cuda:0) driving the HIP path per rank: each rank scans its
position range with lime_fused_dev, the per-rank uint8 tables are summed block-wise (HostComm: the host-staged stand-in
for the RCCL reduce-scatter, which wants one GPU per rank), rank 0 checks the blocks against the oracle.  Also the
bench's N>1 control flow under the same rehearsal backend.  The RCCL calls themselves run on one rank here
(world size 1) -- the 8-GPU node is the driver's."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LIME_ROOT"])
import lime_amd
from lime_amd import dist as ldist
from oracle import oracle_py as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
comm = ldist.HostComm(rank, world, dev); comm.check_uint8_sum_wraps()
n, nr, ng = 900001, 700, 90
lcp, da, eb = O.synth(21, 0, n, nr, ng, 16, 1)
lcp[449000:451000] = 30                                        # a run across the cut between the two ranges
lo, hi, hh = ldist.shard_ranges(n, world)[rank]
ctx = lime_amd.Context(0)
sim_bytes = lime_amd.sim_bytes(nr, ng); blk = ldist.table_block_bytes(sim_bytes, world)
for e in (eb, None):
    tl = torch.from_numpy(lcp[lo:hh].view(np.int32)).to(dev); td = torch.from_numpy(da[lo:hh].view(np.int32)).to(dev)
    te = None if e is None else torch.from_numpy(e[lo:hh]).to(dev)
    sim = torch.zeros(blk * world, dtype=torch.uint8, device=dev); mine = torch.empty(blk, dtype=torch.uint8, device=dev)
    ctx.fused_dev(tl, td, te, hi - lo, hh - lo, hh == n, nr, ng, 16, sim)
    s, rc = ctx.stats(); assert rc == 0, rc
    comm.reduce_scatter_tables(sim, mine, blk)
    nc, ml = comm.combine_counters(int(s.n_clusters), int(s.max_len))
    gathered = [torch.empty(blk, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, mine.cpu())
    if rank == 0:
        cl, enc, eml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, e, cl, nr, ng, threads=4)
        got = torch.cat(gathered)[:nr * ng].numpy().reshape(nr, ng)
        assert (nc, ml) == (enc, eml), ((nc, ml), (enc, eml))
        assert np.array_equal(got, exp), "summed shard tables differ from the oracle"
ctx.close(); dist.destroy_process_group()
if rank == 0: print("DIST_OK")
'''


WORKER_RECORDS = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LIME_ROOT"])
import lime_amd
from lime_amd import dist as ldist
from oracle import oracle_py as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
comm = ldist.HostComm(rank, world, dev)
n, nr, ng = 900001, 7000, 90
lcp, da, eb = O.synth(22, 0, n, nr, ng, 16, 1)
lcp[449000:451000] = 30                                        # a run across the cut between the two ranges: a long cluster
lo, hi, hh = ldist.shard_ranges(n, world)[rank]
ctx = lime_amd.Context(0)
n_bins, bin_shift = ctx.records_layout(nr, ng)
per = (n_bins + world - 1) // world
for e in (eb, None):
    tl = torch.from_numpy(lcp[lo:hh].view(np.int32)).to(dev); td = torch.from_numpy(da[lo:hh].view(np.int32)).to(dev)
    te = None if e is None else torch.from_numpy(e[lo:hh]).to(dev)
    ctx.fused_records_dev(tl, td, te, hi - lo, hh - lo, hh == n, nr, ng, 16)
    s, rc = ctx.stats(); assert rc == 0, rc
    blk = torch.full((per << bin_shift,), 0xCD, dtype=torch.uint8, device=dev)
    cell_lo, nbytes = comm.exchange_records(ctx, nr, ng, blk)
    nc, ml = comm.combine_counters(int(s.n_clusters), int(s.max_len))
    parts = [None] * world
    dist.all_gather_object(parts, (cell_lo, blk[:nbytes].cpu().numpy()))
    if rank == 0:
        cl, enc, eml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, e, cl, nr, ng, threads=4)
        got = np.concatenate([p for _, p in sorted(parts, key=lambda x: x[0])])[:nr * ng].reshape(nr, ng)
        assert (nc, ml) == (enc, eml), ((nc, ml), (enc, eml))
        assert np.array_equal(got, exp), "the owners' blocks differ from the oracle"
ctx.close(); dist.destroy_process_group()
if rank == 0: print("DIST_OK")
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _torchrun(args, env_extra, timeout=600, bench=False):
    env = dict(os.environ, LIME_ROOT=ROOT, **env_extra)
    if bench:                                   # bench.py refuses every LIME_* variable it does not know (the test hooks above all)
        env = {k: v for k, v in env.items() if not k.startswith("LIME_") or k in env_extra}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, capture_output=True, timeout=timeout, env=env, cwd=ROOT)


def test_two_ranks_run_the_hip_path_and_sum_to_the_oracle(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    r = _torchrun([str(w)], {})
    assert r.returncode == 0 and b"DIST_OK" in r.stdout, r.stderr.decode()[-3000:]


def test_two_ranks_exchange_update_records_and_build_their_blocks(tmp_path):
    """owner-partitioned exchange (lime_fused_records_dev -> exchange -> lime_apply_records_dev) with two ranks; the
    transport is the host-staged stand-in, everything else the product path"""
    w = tmp_path / "worker_records.py"
    w.write_text(WORKER_RECORDS)
    r = _torchrun([str(w)], {})
    assert r.returncode == 0 and b"DIST_OK" in r.stdout, r.stderr.decode()[-3000:]


def test_bench_control_flow_with_two_ranks():
    """bench.py --gpus 2 (strong scaling of a small collection, exposed and overlapped exchange) under the rehearsal backend"""
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c2", "--scaling", "strong",
                   "--n-total", "30000000"], {"LIME_BENCH_BACKEND": "gloo"}, bench=True)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["symbols_total"] == 30000000
    # ONE invocation carries all four series: dense exposed = the headline, overlapped, the record exchange, and the probe's pick
    for key in ("overlapped", "sparse_exchange", "auto"):
        e = d["also"][key]
        assert e["value"] > 0 and e["ms_per_step"] > 0 and e["exchange_ms_slowest_rank"] >= 0 and set(e["parts_ms_slowest_rank"]) >= {"scan", "pass"}, (key, e)
    assert d["also"]["sparse_exchange"]["exchange"] == "sparse" and d["also"]["auto"]["exchange"] in ("dense", "sparse")
    assert d["comm"]["world_size"] == 2 and d["comm"]["nccl_comm_count"] == 2 and d["comm"]["uint8_sum_wraps"] is True
    assert d["roofline"]["pass_parts_ms_slowest_rank"]["scan"] > 0
    # the same series with the owner-partitioned exchange of update records
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c2", "--scaling", "strong",
                   "--n-total", "30000000", "--exchange", "sparse", "--no-also"], {"LIME_BENCH_BACKEND": "gloo"}, bench=True)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d2 = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["value"] > 0 and "owner-partitioned" in d2["config"]["sharding"]
    assert d2["config"]["n_clusters"] == d["config"]["n_clusters"]
    # and with the exchange chosen from a probe pass: 14 MB of records against a 50 MB table -> records
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c2", "--scaling", "strong",
                   "--n-total", "30000000", "--exchange", "auto", "--no-also"], {"LIME_BENCH_BACKEND": "gloo"}, bench=True)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d3 = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert "owner-partitioned" in d3["config"]["sharding"] and d3["config"]["n_clusters"] == d["config"]["n_clusters"]


def test_rccl_calls_through_the_c_abi_on_one_rank():
    """lime_comm_* with a world of one rank (the one GPU of this box): id, init, the uint8 reduce-scatter / all-reduce
    self-check and the counter combination"""
    import torch
    import lime_amd
    from lime_amd import dist as ldist
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        comm = ldist.Comm(0, 1, dev)
        comm.check_uint8_sum_wraps()
        assert comm.count() == 1                                     # ncclCommCount through the C ABI
        assert comm.combine_counters(123, 45) == (123, 45)
        src = torch.arange(4096, dtype=torch.int32, device=dev).to(torch.uint8)
        out = torch.zeros(4096, dtype=torch.uint8, device=dev)
        comm.reduce_scatter_tables(src, out, 4096, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(src, out)
        # the owner-partitioned exchange through RCCL (all-gather, send / receive to itself, all-gather of the long clusters' records)
        from oracle import oracle_py as O
        n, nr, ng = 600001, 4000, 70
        lcp, da, eb = O.synth(23, 0, n, nr, ng, 16, 1)
        lcp[300000:300900] = 25
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, eb, cl, nr, ng, threads=4)
        ctx = lime_amd.Context(0)
        tl = torch.from_numpy(lcp.view(np.int32)).to(dev); td = torch.from_numpy(da.view(np.int32)).to(dev); te = torch.from_numpy(eb).to(dev)
        ctx.fused_records_dev(tl, td, te, n, n, True, nr, ng, 16)
        s, rc = ctx.stats(); assert rc == 0
        n_bins, bin_shift = ctx.records_layout(nr, ng)
        blk = torch.full((n_bins << bin_shift,), 0xEE, dtype=torch.uint8, device=dev)
        cell_lo, nbytes = comm.exchange_records(ctx, nr, ng, blk, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert (cell_lo, nbytes) == (0, lime_amd.sim_bytes(nr, ng))
        assert np.array_equal(blk[:nr * ng].cpu().numpy().reshape(nr, ng), exp) and (s.n_clusters, s.max_len) == (nc, ml)
        ctx.close()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_fused_multi_single_device_matches_oracle():
    """lime_fused_multi (one process, n devices) with the one device of this box"""
    import ctypes as C
    import lime_amd
    from lime_amd import _lib
    from oracle import oracle_py as O
    lib = _lib.load()
    n, nr, ng = 500001, 300, 40
    lcp, da, eb = O.synth(5, 0, n, nr, ng, 16, 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        sim = np.zeros((nr, ng), np.uint8)
        gnc, gml = C.c_uint64(0), C.c_uint64(0)
        rc = lib.lime_fused_multi(1, None, lcp.ctypes.data, da.ctypes.data, None if e is None else e.ctypes.data, n, nr, ng, 16,
                                  sim.ctypes.data, C.byref(gnc), C.byref(gml))
        assert rc == 0, lib.lime_comm_error()
        assert (gnc.value, gml.value) == (nc, ml) and np.array_equal(sim, exp)
