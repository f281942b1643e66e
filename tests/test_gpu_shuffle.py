"""GPU: ppo_shuffle_indices (csrc/minibatch.hip) -- the epoch's shuffled index list (reference src/ppo.py:575-576,
np.random.shuffle(indices)).  There is no reference stream to match (numpy's generator is not restated): the
checks are that the output is a permutation, that it equals the oracle's restatement bit for bit, that it is a
function of (seed, stream) only, and -- on the oracle, tests/test_oracle.py -- its statistics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 64, 1000, 4097, 65536, 1 << 20, (1 << 20) + 1, 128 * 65536])
def test_shuffle_is_the_oracles_permutation(n):
    from oracle import cref
    from ppo_amd import ops
    for seed, stream in ((0, 0), (12345, 7), (2**63 + 5, 2**40 + 3)):
        got = ops.shuffle_indices(n, seed, stream).cpu().numpy()
        assert got.dtype == np.int64 and got.shape == (n,)
        if n <= (1 << 20) + 1:
            assert np.array_equal(got, cref.shuffle_indices(n, seed, stream)), (n, seed, stream)
        seen = np.zeros(n, np.uint8)
        seen[got] = 1                                   # raises on an index outside [0, n)
        assert int(seen.sum()) == n, (n, seed, stream)  # every index exactly once


def test_shuffle_depends_on_seed_and_stream_only():
    from ppo_amd import ops
    n = 300001
    a = ops.shuffle_indices(n, 9, 4)
    b = ops.shuffle_indices(n, 9, 4, out=torch.empty(n, dtype=torch.int64, device="cuda"))
    assert torch.equal(a, b)
    for other in (ops.shuffle_indices(n, 9, 5), ops.shuffle_indices(n, 10, 4)):
        same = float((other == a).float().mean())
        assert same < 1e-3, same                        # two random permutations agree in ~1 / n places


def test_minibatches_of_a_shuffled_epoch_mix_the_rollout():
    """BASELINE configs[1]: 128 x 65 536 rows, four minibatches.  Each minibatch draws from every time step and every
    env block about equally (a [T, N] rollout flattened row-major: row = t N + n)."""
    from ppo_amd import ops
    T, N = 128, 65536
    idx = ops.shuffle_indices(T * N, 3, 0)
    for mb in idx.view(4, -1):
        t = torch.bincount(mb // N, minlength=T).double()            # expected 16384 per time step, sd ~ 111
        assert float((t - 16384).abs().max()) < 6 * 111, float((t - 16384).abs().max())
        blk = torch.bincount((mb % N) // 1024, minlength=64).double()   # expected 32768 per block of 1024 envs, sd ~ 157
        assert float((blk - 32768).abs().max()) < 6 * 157


def test_shuffle_rejects_bad_arguments():
    from ppo_amd import _lib
    lib = _lib.load()
    out = torch.empty(8, dtype=torch.int64, device="cuda")
    assert lib.ppo_shuffle_indices(8, 1, 1, out.data_ptr(), None) == 0
    assert lib.ppo_shuffle_indices(0, 1, 1, None, None) == 0
    assert lib.ppo_shuffle_indices(8, 1, 1, None, None) == 1
    assert lib.ppo_shuffle_indices(-1, 1, 1, out.data_ptr(), None) == 1
    assert lib.ppo_shuffle_indices((1 << 32) + 1, 1, 1, out.data_ptr(), None) == 1
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,batch", [(8, 3), (1000, 256), (4097, 4097), (4097, 1025), (65536, 8192), (100003, 20000),
                                     (1 << 20, 1 << 18), (8388608, 2097152), (8388609, 1048577)])
def test_epoch_advantage_statistics_follow_the_permutation(n, batch):
    """ppo_adv_stats_epoch (the shuffle's bijection run backwards, one streaming pass) against ppo_adv_stats_gather
    on the slices of the permutation ppo_shuffle_indices writes, and against numpy in float64: counts exact, mean
    and M2 to f64 summation-order accuracy.  A wrong inverse would put values into other minibatches: the offsets
    make every minibatch's mean different by far more than the tolerance."""
    from ppo_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(n)
    adv = torch.randn(n, device="cuda", generator=gen) * 3.0 + 40.0 + torch.arange(n, device="cuda") * (7.0 / n)
    for seed, stream in ((5, 0), (2**63 + 11, (3 << 40) + 17)):
        perm = ops.shuffle_indices(n, seed, stream)
        got = ops.adv_stats_epoch(adv, seed, stream, batch).cpu().numpy()
        num_mb = (n + batch - 1) // batch
        assert got.shape == (num_mb, 3)
        a64 = adv.double().cpu().numpy()
        p = perm.cpu().numpy()
        for m in range(num_mb):
            idx = perm[m * batch:(m + 1) * batch].contiguous()
            ref = ops.adv_stats(adv, idx).cpu().numpy()
            x = a64[p[m * batch:(m + 1) * batch]]
            assert got[m, 0] == ref[0] == len(x)
            assert abs(got[m, 1] - ref[1]) <= 1e-12 * abs(ref[1]) and abs(got[m, 1] - x.mean()) <= 1e-11 * abs(x.mean())
            m2 = ((x - x.mean()) ** 2).sum()
            assert abs(got[m, 2] - ref[2]) <= 1e-9 * max(ref[2], 1e-30) and abs(got[m, 2] - m2) <= 1e-9 * max(m2, 1e-30)
        again = ops.adv_stats_epoch(adv, seed, stream, batch).cpu().numpy()
        assert (again == got).all()                              # fixed summation order


def test_epoch_advantage_statistics_argument_checks():
    from ppo_amd import _lib
    lib = _lib.load()
    adv = torch.randn(1024, device="cuda")
    out = torch.empty((8, 3), dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.ppo_adv_stats_epoch_workspace_bytes(), dtype=torch.uint8, device="cuda")
    call = lambda n=1024, batch=256, ws_bytes=ws.numel(), o=out.data_ptr(): lib.ppo_adv_stats_epoch(
        adv.data_ptr(), n, 1, 2, batch, o, ws.data_ptr(), ws_bytes, None)
    assert call() == 0
    for bad in (dict(n=0), dict(batch=0), dict(batch=100), dict(ws_bytes=64), dict(o=None)):   # batch 100: 11 minibatches > 8
        assert call(**bad) == 1, bad
    torch.cuda.synchronize()
