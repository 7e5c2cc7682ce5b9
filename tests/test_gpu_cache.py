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
