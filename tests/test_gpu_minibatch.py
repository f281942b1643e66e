"""GPU parity: ppo_minibatch_gather (through the C ABI) vs numpy fancy indexing -- the
reference's minibatch construction at src/ppo.py:582-588.  Pure data movement: bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(M, B, state_shape, act_shape, act_dtype, seed, negative=False):
    from ppo_amd import ops
    rng = np.random.default_rng(seed)
    states = rng.standard_normal((M, *state_shape)).astype(np.float32)
    if act_dtype == np.int32:
        actions = rng.integers(0, 5, (M, *act_shape)).astype(np.int32)
    else:
        actions = rng.standard_normal((M, *act_shape)).astype(np.float32)
    lp, v, adv = (rng.standard_normal(M).astype(np.float32) for _ in range(3))
    idx = rng.permutation(M)[:B].astype(np.int64)
    if negative:
        idx[::3] -= M
    got = ops.minibatch_gather(torch.from_numpy(idx).cuda(), *(torch.from_numpy(x).cuda()
                               for x in (states, actions, lp, v, adv)), check_index=True)
    for g, src in zip(got, (states, actions, lp, v, adv)):
        want = src[idx]
        g = g.cpu().numpy()
        assert g.dtype == want.dtype and g.shape == want.shape
        assert np.array_equal(g.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("state_shape,act_shape,act_dtype", [
    ((4,), (), np.int32),        # CartPole: 16-byte observation rows (vector path), Discrete actions
    ((6,), (), np.int32),        # Acrobot: 24-byte rows (word path)
    ((27,), (8,), np.float32),   # Ant-shaped: Box actions
    ((5,), (4,), np.float32),    # 4-wide Box action rows take the vector path
    ((3, 2, 2), (), np.int32),   # image-like observation
])
@pytest.mark.parametrize("M,B", [(1, 1), (1000, 1000), (4097, 63), (1 << 16, 1 << 14)])
def test_gather_equals_fancy_indexing(state_shape, act_shape, act_dtype, M, B):
    _case(M, B, state_shape, act_shape, act_dtype, seed=M + B)


@pytest.mark.parametrize("state_shape", [(84, 84, 4), (8, 4), (33,), (9, 4)])
def test_wide_observation_rows(state_shape):
    """Image observations (BASELINE configs[3]: 84 x 84 x 4 frames): rows of >= 32 words that are
    16-byte granular move through the wide-row kernel; 33-word rows stay on the per-sample path."""
    _case(300, 170, state_shape, (), np.int32, seed=len(state_shape))
    _case(300, 170, state_shape, (), np.int32, seed=7, negative=True)
    from ppo_amd import ops
    M = 6
    z = torch.zeros(M, device="cuda")
    args = (torch.zeros(M, *state_shape, device="cuda"), torch.zeros(M, dtype=torch.int32, device="cuda"), z, z, z)
    with pytest.raises(IndexError):
        ops.minibatch_gather(torch.tensor([2, 6], device="cuda"), *args, check_index=True)


def test_negative_indices_wrap_and_out_of_range_raises():
    from ppo_amd import ops
    _case(500, 200, (4,), (), np.int32, seed=1, negative=True)
    M = 10
    z = torch.zeros(M, device="cuda")
    args = (torch.zeros(M, 4, device="cuda"), torch.zeros(M, dtype=torch.int32, device="cuda"), z, z, z)
    with pytest.raises(IndexError):
        ops.minibatch_gather(torch.tensor([0, 10], device="cuda"), *args, check_index=True)
    with pytest.raises(IndexError):
        ops.minibatch_gather(torch.tensor([-11], device="cuda"), *args, check_index=True)


def test_full_scale_permutation_round_trip():
    """C2 scale (M = 65536 x 128): gathering with a permutation and then with its inverse is the
    identity -- a size-independent property, no oracle needed."""
    from ppo_amd import ops
    M = 65536 * 128
    g = torch.Generator(device="cuda").manual_seed(0)
    states = torch.randn(M, 4, device="cuda", generator=g)
    actions = torch.randint(0, 2, (M,), device="cuda", dtype=torch.int32, generator=g)
    lp, v, adv = (torch.randn(M, device="cuda", generator=g) for _ in range(3))
    perm = torch.randperm(M, device="cuda", generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(M, device="cuda")
    once = ops.minibatch_gather(perm, states, actions, lp, v, adv)
    back = ops.minibatch_gather(inv, *once)
    for b, src in zip(back, (states, actions, lp, v, adv)):
        assert torch.equal(b, src)
