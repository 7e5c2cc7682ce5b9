"""The library's process-wide cache of device blocks (lime_trim_cache, include/lime_hip.h): contexts of one process reuse each other's large blocks
instead of returning them to the driver, which clears recycled pages inside hipMalloc at about 30 GB/s.  Under LIME_TEST_HOOKS
(tests/conftest.py) every recycled block is filled with 0xA5 first: a pass on recycled blocks must give what a pass on fresh ones gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pass(n, nr, ng, mode):
    import torch
    import lime_amd
    dev = torch.device("cuda", 0)
    c = lime_amd.Context()
    try:
        c.set_option("update_path", "bin")
        lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp); eb = torch.empty(n, dtype=torch.uint8, device=dev)
        c.synth_dev(7, 0, n, nr, ng, 16, mode, lcp, da, eb)
        sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
        c.fused_dev(lcp, da, eb, n, n, True, nr, ng, 16, sim)
        s, rc = c.stats()
        assert rc == 0 and s.wave_records_max > 0                     # the binned path: record pool + binned records, a few hundred MB
        return sim.cpu().numpy(), (s.n_clusters, s.max_len, s.n_updates), c.host_times()["alloc_ms"]
    finally:
        c.close()


def test_contexts_reuse_released_blocks_and_trim_returns_them():
    import lime_amd
    lime_amd.trim_cache()
    n, nr, ng = 60_000_000, 200_000, 900
    a, ca, _ = _pass(n, nr, ng, 1)
    held = lime_amd.trim_cache()
    assert held >= 64 << 20, held                                     # the closed context left its large blocks with the library ...
    assert lime_amd.trim_cache() == 0                                 # ... and a trim gives all of them back
    b, cb, _ = _pass(n, nr, ng, 1)                                    # fresh blocks again
    c, cc, _ = _pass(n, nr, ng, 1)                                    # recycled (and poisoned) blocks
    d, cd, _ = _pass(n // 2, nr, ng, 0)                               # a smaller pass served from larger cached blocks
    e, ce, _ = _pass(n // 2, nr, ng, 0)
    assert ca == cb == cc and np.array_equal(a, b) and np.array_equal(b, c)
    assert cd == ce and np.array_equal(d, e)
    assert lime_amd.trim_cache() > 0


def test_reserved_block_serves_the_passes_and_goes_back_with_a_trim():
    """lime_reserve: one block taken from the driver at start-up; the large buffers of every later context are carved from it (and come back to it), so
    no pass waits for the driver's allocator: the free memory the driver reports does not move while passes run, results are those of fresh blocks,
    pieces are reused after a context closes (poisoned under the hooks), a second reservation is used when the first is full, and a trim returns
    reserved blocks only when nothing is carved from them."""
    import torch
    import lime_amd
    lime_amd.trim_cache()
    n, nr, ng = 60_000_000, 200_000, 900
    ref, cref, _ = _pass(n, nr, ng, 1)                                # blocks from the driver
    lime_amd.trim_cache()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    lime_amd.reserve(3 << 30)
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 >= (3 << 30)
    dev = torch.device("cuda", 0)
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp); eb = torch.empty(n, dtype=torch.uint8, device=dev)
    sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    c = lime_amd.Context()
    try:
        c.set_option("update_path", "bin")
        c.synth_dev(7, 0, n, nr, ng, 16, 1, lcp, da, eb)
        c.fused_dev(lcp, da, eb, n, n, True, nr, ng, 16, sim)
        s, rc = c.stats()
        assert rc == 0 and (s.n_clusters, s.max_len, s.n_updates) == cref
        assert np.array_equal(sim.cpu().numpy(), ref)
        # only the context's small blocks (below 64 MB each) came from the driver
        assert free2 - torch.cuda.mem_get_info()[0] < (256 << 20), (free2, torch.cuda.mem_get_info()[0])
        assert lime_amd.trim_cache() == 0                              # the reserved block is in use: it stays
        c2 = lime_amd.Context()                                        # a second context beside the first: the rest of the block, then a second reservation
        try:
            c2.set_option("update_path", "bin")
            lime_amd.reserve(1 << 30)
            sim2 = torch.empty_like(sim)
            c2.fused_dev(lcp, da, eb, n, n, True, nr, ng, 16, sim2)
            s2, rc2 = c2.stats()
            assert rc2 == 0 and (s2.n_clusters, s2.max_len, s2.n_updates) == cref and torch.equal(sim, sim2)
        finally:
            c2.close()
    finally:
        c.close()
    again, cagain, _ = _pass(n, nr, ng, 1)                            # pieces that went back are carved again (and poisoned first)
    assert cagain == cref and np.array_equal(again, ref)
    assert lime_amd.trim_cache() >= (4 << 30)                         # nothing carved: both reservations go back to the driver
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free2 + (3 << 30) - (512 << 20)       # (free2: with the first reservation held and this test's arrays allocated)
