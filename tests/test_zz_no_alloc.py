"""GPU: ppo_gae_ragged_noalloc (scratch in a device array inside the library, no allocation) against
ppo_gae_ragged (caller workspace) -- bit for bit, on two calls in a row, the second with a smaller
input, so that stale scratch contents of the first call would show.  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _case(rng, T, N):
    r, v = (rng.standard_normal((T, N)).astype(np.float32) for _ in range(2))
    d = (rng.random((T, N)) < 0.04).astype(np.int32)
    tr = ((rng.random((T, N)) < 0.02) & (d == 0)).astype(np.int32)
    tr[-1] = 1 - d[-1]                                   # forced truncation of the last row (src/ppo.py:352-356)
    E = int((d + tr - d * tr).sum())
    ev = rng.standard_normal(E).astype(np.float32)
    return r, v, ev, d, tr


def _run(lib, fn_noalloc, arrs, gamma=0.99, lam=0.95):
    r, v, ev, d, tr = (torch.from_numpy(x).cuda() for x in arrs)
    T, N = r.shape
    adv = torch.full((T, N), float("nan"), device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if fn_noalloc:
        rc = lib.ppo_gae_ragged_noalloc(r.data_ptr(), v.data_ptr(), ev.data_ptr(), ev.numel(), d.data_ptr(),
                                        tr.data_ptr(), gamma, lam, adv.data_ptr(), T, N, err.data_ptr(), stream)
    else:
        ws = torch.empty(lib.ppo_gae_ragged_workspace_bytes(N), dtype=torch.uint8, device="cuda")
        rc = lib.ppo_gae_ragged(r.data_ptr(), v.data_ptr(), ev.data_ptr(), ev.numel(), d.data_ptr(), tr.data_ptr(),
                                gamma, lam, adv.data_ptr(), T, N, ws.data_ptr(), ws.numel(), err.data_ptr(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    return adv.cpu().numpy()


def test_noalloc_ragged_gae_equals_workspace_form_on_two_calls():
    from ppo_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    big = _case(rng, 96, 5000)          # 20 column blocks of 256, the last one partial
    small = _case(rng, 33, 777)         # fewer columns, fewer blocks: reads a prefix of the same scratch
    for arrs in (big, small):           # two calls in a row on the same GPU's scratch array
        got = _run(lib, True, arrs)
        want = _run(lib, False, arrs)
        assert np.array_equal(got, want)
        assert np.array_equal(got, cref.gae_ragged(*arrs, 0.99, 0.95))   # and both equal the oracle, bit for bit


def test_noalloc_rejects_more_columns_than_its_scratch_holds():
    from ppo_amd import _lib
    lib = _lib.load()
    N = 262144 + 256                    # PPO_GAE_RAGGED_NOALLOC_MAX_N + one block
    z = torch.zeros(N, device="cuda")
    zi = torch.zeros(N, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.ppo_gae_ragged_noalloc(z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, zi.data_ptr(), zi.data_ptr(), 0.99,
                                    0.95, z.data_ptr(), 1, N, err.data_ptr(), None)
    assert rc == 1                      # hipErrorInvalidValue, nothing launched
    # the maximum itself is accepted
    N = 262144
    adv = torch.empty(N, device="cuda")
    one = torch.ones(N, dtype=torch.int32, device="cuda")
    rc = lib.ppo_gae_ragged_noalloc(z.data_ptr(), z.data_ptr(), z.data_ptr(), N, zi.data_ptr(), one.data_ptr(), 0.99,
                                    0.95, adv.data_ptr(), 1, N, err.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(err.item()) == 0 and float(adv.abs().max()) == 0.0


def test_noalloc_wrapper_matches():
    from ppo_amd import ops
    rng = np.random.default_rng(5)
    arrs = _case(rng, 20, 64)
    t = [torch.from_numpy(x).cuda() for x in arrs]
    a = ops.gae_ragged(*t, 0.9, 0.8, noalloc=True)
    b = ops.gae_ragged(*t, 0.9, 0.8)
    assert torch.equal(a, b)
