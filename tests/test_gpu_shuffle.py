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
