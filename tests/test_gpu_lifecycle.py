"""Create / use / destroy cycles: device memory comes back, nothing hangs, objects can go in any order."""
import gc

import numpy as np
import pytest

from tests.conftest import make_tables
from tests.test_gpu_lookup import _bits, _mk

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


@pytest.mark.parametrize("direct", [False, True], ids=["host_gather", "ps_direct_access"])
def test_create_destroy_cycles_release_device_memory(direct):
    from hugectr_backend_amd import hps
    from hugectr_backend_amd.dense import DenseInteraction
    from oracle import hps_oracle as O
    import torch
    tables = make_tables([(20000, 128), (20000, 64)])
    q = np.concatenate([tables[0][0][:3000], tables[1][0][:3000]]).astype(np.int64)
    ref = O.np_lookup(tables, q, [3000, 3000], [0.0, 0.0])
    torch.zeros(1, device="cuda")
    gc.collect()
    base = None
    for cycle in range(6):
        ps, cache, s = _mk(f"life{cycle}{int(direct)}", tables, maxcat=[1, 1], gpucacheper=0.5, max_batch=4096,
                           hit_rate_threshold=0.9 if cycle % 2 else 1.0, extra={"ps_direct_access": direct})
        s2 = hps.LookupSession.create(ps, f"life{cycle}{int(direct)}", cache)
        for sess in (s, s2):
            out = sess.lookup(q, [3000, 3000]).cpu().numpy()
            if cycle % 2 == 0:
                assert np.array_equal(_bits(out), _bits(ref))
        op = DenseInteraction([np.zeros((13, 64), np.float32), np.zeros((64, 128), np.float32)],
                              [np.zeros(64, np.float32), np.zeros(128, np.float32)], 1, 128)
        # tear down in a different order every cycle
        objs = [s, s2, cache, ps, op]
        order = np.random.default_rng(cycle).permutation(len(objs))
        cache.wait_async()
        for i in order:
            o = objs[i]
            if hasattr(o, "close"):
                o.close()
        del s, s2, cache, ps, op, objs, out
        gc.collect()
        free = _free_bytes()
        if cycle == 1:
            base = free          # after the first full cycle the allocator pools (torch, HIP) have settled
        elif cycle > 1:
            assert base - free < (64 << 20), f"cycle {cycle}: {(base - free) >> 20} MiB of device memory not returned"
