"""Tensor-level wrappers over the C ABI (include/ppo_amd.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; all arithmetic on the
hot path happens in libppo_amd.so.  Tensors that live on the CPU are staged to the HIP
device and the result is staged back -- the computation itself never runs on the CPU, and
with no HIP device every op raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import os

import torch

from . import _lib

_ws_cache: dict = {}
_GAE_VARIANT = int(os.environ["PPO_GAE_VARIANT"]) if "PPO_GAE_VARIANT" in os.environ else None   # tuning only


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.PpoAmdError("ppo_amd needs a HIP device (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    # torch.cuda.current_stream().cuda_stream by its two C calls (the Python wrapper objects cost ~9 us per launch)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _dev(t: torch.Tensor, dtype) -> torch.Tensor:
    """Contiguous tensor of `dtype` on the HIP device (a copy only if needed)."""
    return t.detach().to(device=device(), dtype=dtype).contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _workspace(nbytes: int, dev: torch.device) -> torch.Tensor:
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws


# --------------------------------------------------------------------------- GAE
def gae_dense(rewards, values, end_value_dense, done_flags, trunc_flags, gamma, lam,
              want_returns=False, out=None, variant=None, prof_slot=None):
    """ppo_gae_dense: replaces reference src/ppo.py:452-488 with dense bootstrap values.
    prof_slot (bench only): time the dispatch into that event slot (ppo_gae_dense_timed)."""
    lib = _lib.load()
    src_dev = rewards.device
    r, v, ev = _dev(rewards, torch.float32), _dev(values, torch.float32), _dev(end_value_dense, torch.float32)
    d, tr = _dev(done_flags, torch.int32), _dev(trunc_flags, torch.int32)
    T, N = r.shape
    assert v.shape == (T, N) and ev.shape == (T, N) and d.shape == (T, N) and tr.shape == (T, N)
    adv = out if out is not None else torch.empty((T, N), dtype=torch.float32, device=r.device)
    ret = torch.empty((T, N), dtype=torch.float32, device=r.device) if want_returns else None
    if variant is None and _GAE_VARIANT is not None:
        variant = _GAE_VARIANT
    if prof_slot is not None:
        rc = lib.ppo_gae_dense_timed(_ptr(r), _ptr(v), _ptr(ev), _ptr(d), _ptr(tr), float(gamma), float(lam),
                                     _ptr(adv), _ptr(ret), T, N, int(prof_slot), _stream())
    elif variant is None:
        rc = lib.ppo_gae_dense(_ptr(r), _ptr(v), _ptr(ev), _ptr(d), _ptr(tr), float(gamma), float(lam),
                               _ptr(adv), _ptr(ret), T, N, _stream())
    else:
        rc = lib.ppo_gae_dense_variant(_ptr(r), _ptr(v), _ptr(ev), _ptr(d), _ptr(tr), float(gamma), float(lam),
                                       _ptr(adv), _ptr(ret), T, N, int(variant), _stream())
    _lib.check(rc, "ppo_gae_dense")
    if src_dev.type != "cuda":
        adv = adv.to(src_dev)
        ret = ret.to(src_dev) if ret is not None else None
    return (adv, ret) if want_returns else adv


def gae_flag_words(T: int, N: int, device) -> torch.Tensor:
    """The [ceil(T/16), N] u32 (held as int32) flag-word buffer of ppo_rollout_fused_packed / ppo_gae_dense_packed."""
    n = _lib.load().ppo_gae_flag_words(int(T), int(N))
    return torch.empty((n // max(N, 1), N), dtype=torch.int32, device=device)


def pack_flag_words(done_flags, trunc_flags) -> torch.Tensor:
    """[T,N] i32 done / trunc (values 0 / 1) -> flag words, on the tensors' device (tests and tools: the rollout
    kernel writes the words itself)."""
    T, N = done_flags.shape
    code = (done_flags.to(torch.int64) | (trunc_flags.to(torch.int64) << 1))
    rows = (T + 15) // 16 * 16
    if rows != T:
        code = torch.cat([code, torch.zeros((rows - T, N), dtype=torch.int64, device=code.device)])
    shifts = (2 * torch.arange(16, device=code.device, dtype=torch.int64)).view(1, 16, 1)
    words = (code.view(rows // 16, 16, N) << shifts).sum(1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()   # the u32 bit pattern


def gae_dense_packed(rewards, values, end_value_dense, flag_words, gamma, lam, want_returns=False, out=None, variant=None,
                     prof_slot=None):
    """ppo_gae_dense_packed: the scan of reference src/ppo.py:452-488 reading the two flag bits per element the
    whole-rollout kernel leaves; device tensors only.  prof_slot (bench only): time the dispatch into that slot."""
    lib = _lib.load()
    T, N = rewards.shape
    for x in (rewards, values, end_value_dense):
        assert x.is_cuda and x.dtype == torch.float32 and x.shape == (T, N) and x.is_contiguous(), (x.dtype, x.shape)
    assert flag_words.is_cuda and flag_words.dtype == torch.int32 and flag_words.is_contiguous()
    assert flag_words.shape == ((T + 15) // 16, N), (flag_words.shape, T, N)
    adv = out if out is not None else torch.empty((T, N), dtype=torch.float32, device=rewards.device)
    ret = torch.empty((T, N), dtype=torch.float32, device=rewards.device) if want_returns else None
    args = (_ptr(rewards), _ptr(values), _ptr(end_value_dense), _ptr(flag_words), float(gamma), float(lam), _ptr(adv),
            _ptr(ret), T, N)
    if prof_slot is not None:
        _lib.check(lib.ppo_prof_arm(int(prof_slot), 1), "ppo_prof_arm")
    if variant is None:
        rc = lib.ppo_gae_dense_packed(*args, _stream())
    else:
        rc = lib.ppo_gae_dense_packed_variant(*args, int(variant), _stream())
    if prof_slot is not None:
        lib.ppo_prof_arm(int(prof_slot), 0)     # (an argument check may have returned before the launch)
    _lib.check(rc, "ppo_gae_dense_packed")
    return (adv, ret) if want_returns else adv


def gae_ragged(rewards, values, end_values, done_flags, trunc_flags, gamma, lam, check_index=True, noalloc=False,
               out=None, err=None):
    """ppo_gae_ragged: exact drop-in for PPO.compute_advantages (reference src/ppo.py:452-488).
    noalloc=True goes through ppo_gae_ragged_noalloc (scratch inside the library, no workspace); with
    `out` [T,N] f32 and `err` [1] i32 device tensors supplied and device inputs of the right dtype,
    that path allocates nothing at all."""
    lib = _lib.load()
    src_dev = rewards.device
    r, v, ev = _dev(rewards, torch.float32), _dev(values, torch.float32), _dev(end_values, torch.float32)
    d, tr = _dev(done_flags, torch.int32), _dev(trunc_flags, torch.int32)
    T, N = r.shape
    assert v.shape == (T, N) and d.shape == (T, N) and tr.shape == (T, N) and ev.dim() == 1
    adv = out if out is not None else torch.empty((T, N), dtype=torch.float32, device=r.device)
    if err is None:
        err = torch.zeros(1, dtype=torch.int32, device=r.device)
    if noalloc:
        rc = lib.ppo_gae_ragged_noalloc(_ptr(r), _ptr(v), _ptr(ev), ev.numel(), _ptr(d), _ptr(tr), float(gamma),
                                        float(lam), _ptr(adv), T, N, _ptr(err), _stream())
        _lib.check(rc, "ppo_gae_ragged_noalloc")
    else:
        nws = lib.ppo_gae_ragged_workspace_bytes(N)
        ws = torch.empty(max(nws, 8), dtype=torch.uint8, device=r.device)
        rc = lib.ppo_gae_ragged(_ptr(r), _ptr(v), _ptr(ev), ev.numel(), _ptr(d), _ptr(tr), float(gamma), float(lam),
                                _ptr(adv), T, N, _ptr(ws), ws.numel(), _ptr(err), _stream())
        _lib.check(rc, "ppo_gae_ragged")
    if check_index and int(err.item()) != 0:  # torch raises IndexError at src/ppo.py:478
        raise IndexError("index out of range in end_values (done+trunc bookkeeping does not match end_values)")
    return adv if src_dev.type == "cuda" else adv.to(src_dev)


# --------------------------------------------------------------------------- loss
def adv_stats(adv: torch.Tensor, idx: torch.Tensor | None = None) -> torch.Tensor:
    """{count, mean, M2} (fp64, device) of a flat advantage tensor (reference src/ppo.py:501); with idx (device
    int64), of adv[idx] without gathering it (ppo_adv_stats_gather: the same numbers bit for bit)."""
    lib = _lib.load()
    a = _dev(adv, torch.float32).reshape(-1)
    stats = torch.empty(3, dtype=torch.float64, device=a.device)
    n = a.numel() if idx is None else idx.numel()
    ws = _workspace(lib.ppo_loss_workspace_bytes(n), a.device)
    if idx is None:
        _lib.check(lib.ppo_adv_stats(_ptr(a), n, _ptr(stats), _ptr(ws), ws.numel(), _stream()), "ppo_adv_stats")
    else:
        assert idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous() and idx.dim() == 1
        _lib.check(lib.ppo_adv_stats_gather(_ptr(a), _ptr(idx), n, _ptr(stats), _ptr(ws), ws.numel(), _stream()),
                   "ppo_adv_stats_gather")
    return stats


def merge_stats(stats_list):
    """Chan et al. merge of {count, mean, M2} triples (data-parallel ranks)."""
    n, mean, m2 = stats_list[0]
    for nb, mb, m2b in stats_list[1:]:
        tot = n + nb
        delta = mb - mean
        mean = mean + delta * nb / tot
        m2 = m2 + m2b + delta * delta * n * nb / tot
        n = tot
    return torch.stack([n, mean, m2])


def loss_fwd_bwd(x, old_lp, new_v, old_v, adv, clip, norm_adv, clip_va, ratio_input, stats3=None,
                 want_grads=True):
    """ppo_loss_fwd_bwd: returns (out4 [device f32x4], g_x, g_v)."""
    lib = _lib.load()
    x_, nv, ov, a = (_dev(t, torch.float32).reshape(-1) for t in (x, new_v, old_v, adv))
    olp = None if old_lp is None else _dev(old_lp, torch.float32).reshape(-1)
    B = x_.numel()
    assert nv.numel() == B and ov.numel() == B and a.numel() == B
    if norm_adv and stats3 is None:
        stats3 = adv_stats(a)
    out4 = torch.empty(4, dtype=torch.float32, device=x_.device)
    g_x = torch.empty(B, dtype=torch.float32, device=x_.device) if want_grads else None
    g_v = torch.empty(B, dtype=torch.float32, device=x_.device) if want_grads else None
    ws = _workspace(lib.ppo_loss_workspace_bytes(B), x_.device)
    rc = lib.ppo_loss_fwd_bwd(_ptr(x_), _ptr(olp), _ptr(nv), _ptr(ov), _ptr(a), _ptr(stats3), float(clip),
                              int(bool(norm_adv)), int(bool(clip_va)), int(bool(ratio_input)), B,
                              _ptr(out4), _ptr(g_x), _ptr(g_v), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "ppo_loss_fwd_bwd")
    return out4, g_x, g_v


class _AttachGrad(torch.autograd.Function):
    """value(inp) with d value / d inp = grad, both precomputed by ppo_loss_fwd_bwd.

    Policy loss and value loss get one node EACH (fed by the same kernel launch): with split
    networks the reference back-propagates the two losses separately and steps the policy
    optimiser in between (reference src/ppo.py:614-625), so the value loss must not be connected
    to the policy branch of the graph."""

    @staticmethod
    def forward(ctx, inp, value, grad):
        ctx.save_for_backward(grad)
        ctx.meta = (inp.shape, inp.device)
        return value.clone()

    @staticmethod
    def backward(ctx, g_out):
        (grad,) = ctx.saved_tensors
        shape, dev = ctx.meta
        return (grad * g_out.to(grad.device)).to(dev).reshape(shape), None, None


def ppo_losses(x, new_v, old_lp, old_v, adv, clip, norm_adv, clip_va, ratio_input, stats3=None):
    """(policy_loss, value_loss, clip_frac, kl_div), each 0-d f32 on x.device; the first two are
    differentiable w.r.t. x (ratios or new log-probs) and new_v respectively."""
    need = torch.is_grad_enabled() and (x.requires_grad or new_v.requires_grad)
    with torch.no_grad():
        out4, g_x, g_v = loss_fwd_bwd(x, old_lp, new_v, old_v, adv, float(clip), bool(norm_adv), bool(clip_va),
                                      bool(ratio_input), stats3=stats3, want_grads=need)
        o = out4.to(x.device)
    pl, vl, cf, kl = o[0], o[1], o[2], o[3]
    if need and x.requires_grad:
        pl = _AttachGrad.apply(x, pl, g_x)
    if need and new_v.requires_grad:
        vl = _AttachGrad.apply(new_v, vl, g_v)
    return pl, vl, cf, kl


def loss_logits_fwd_bwd(logits, actions, old_lp, new_v, old_v, adv, clip, norm_adv, clip_va, stats3=None,
                        want_grads=True):
    """ppo_loss_logits_fwd_bwd: returns (out5 [device f32x5], g_logits_pl, g_logits_ent, g_v)."""
    lib = _lib.load()
    lg = _dev(logits, torch.float32)
    assert lg.dim() == 2, "logits must be [B, num_actions]"
    B, A = lg.shape
    act = _dev(actions, torch.int32).reshape(-1)
    olp, nv, ov, a = (_dev(t, torch.float32).reshape(-1) for t in (old_lp, new_v, old_v, adv))
    assert act.numel() == B and olp.numel() == B and nv.numel() == B and ov.numel() == B and a.numel() == B
    if norm_adv and stats3 is None:
        stats3 = adv_stats(a)
    dev = lg.device
    out5 = torch.empty(5, dtype=torch.float32, device=dev)
    g_pl = torch.empty(B, A, dtype=torch.float32, device=dev) if want_grads else None
    g_ent = torch.empty(B, A, dtype=torch.float32, device=dev) if want_grads else None
    g_v = torch.empty(B, dtype=torch.float32, device=dev) if want_grads else None
    ws = _workspace(lib.ppo_loss_workspace_bytes(B), dev)
    rc = lib.ppo_loss_logits_fwd_bwd(_ptr(lg), A, _ptr(act), _ptr(olp), _ptr(nv), _ptr(ov), _ptr(a), _ptr(stats3),
                                     float(clip), int(bool(norm_adv)), int(bool(clip_va)), B, _ptr(out5),
                                     _ptr(g_pl), _ptr(g_ent), _ptr(g_v), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "ppo_loss_logits_fwd_bwd")
    return out5, g_pl, g_ent, g_v


def ppo_losses_discrete(logits, actions, new_v, old_lp, old_v, adv, clip, norm_adv, clip_va, stats3=None):
    """(policy_loss, value_loss, clip_frac, kl_div, entropy) for a Categorical head given its raw
    logits; policy_loss and entropy are differentiable w.r.t. logits, value_loss w.r.t. new_v.
    One kernel pass replaces Categorical(logits).log_prob / .entropy().mean() (reference
    src/ppo.py:186-189,212-214), the ratio (:595) and compute_losses (:490-527)."""
    need = torch.is_grad_enabled() and (logits.requires_grad or new_v.requires_grad)
    with torch.no_grad():
        out5, g_pl, g_ent, g_v = loss_logits_fwd_bwd(logits, actions, old_lp, new_v, old_v, adv, float(clip),
                                                     bool(norm_adv), bool(clip_va), stats3=stats3, want_grads=need)
        o = out5.to(logits.device)
    pl, vl, cf, kl, ent = o[0], o[1], o[2], o[3], o[4]
    if need and logits.requires_grad:
        pl = _AttachGrad.apply(logits, pl, g_pl)
        ent = _AttachGrad.apply(logits, ent, g_ent)
    if need and new_v.requires_grad:
        vl = _AttachGrad.apply(new_v, vl, g_v)
    return pl, vl, cf, kl, ent


# --------------------------------------------------------------------------- minibatch
def shuffle_indices(n: int, seed: int, stream_id: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """ppo_shuffle_indices: a pseudo-random permutation of 0 .. n-1 as a device int64 tensor, a function of
    (seed, stream_id) only (reference src/ppo.py:575-576: np.random.shuffle(indices), once per epoch)."""
    lib = _lib.load()
    dev = device()
    if out is None:
        out = torch.empty(n, dtype=torch.int64, device=dev)
    assert out.is_cuda and out.dtype == torch.int64 and out.numel() == n and out.is_contiguous()
    rc = lib.ppo_shuffle_indices(int(n), int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1), _ptr(out), _stream())
    _lib.check(rc, "ppo_shuffle_indices")
    return out


ADV_STATS_EPOCH_MAX_MB = 8
_epoch_ws = {}


def adv_stats_epoch(adv: torch.Tensor, seed: int, stream_id: int, batch: int) -> torch.Tensor:
    """ppo_adv_stats_epoch: [num_mb, 3] f64 {count, mean, M2} of adv[perm[m * batch:(m + 1) * batch]] for every
    minibatch m of the epoch whose permutation is shuffle_indices(adv.numel(), seed, stream_id) -- one coalesced
    pass over adv instead of one gather per minibatch (reference src/ppo.py:578-588, :501)."""
    lib = _lib.load()
    a = adv.reshape(-1)
    assert a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()
    n = a.numel()
    num_mb = (n + batch - 1) // batch
    assert 1 <= num_mb <= ADV_STATS_EPOCH_MAX_MB
    ws = _epoch_ws.get(a.device)
    if ws is None:
        ws = _epoch_ws[a.device] = torch.empty(lib.ppo_adv_stats_epoch_workspace_bytes(), dtype=torch.uint8, device=a.device)
    out = torch.empty((num_mb, 3), dtype=torch.float64, device=a.device)
    rc = lib.ppo_adv_stats_epoch(_ptr(a), n, int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1), int(batch), _ptr(out),
                                 _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "ppo_adv_stats_epoch")
    return out


def minibatch_gather(idx, states, actions, log_probs, values, adv, check_index=False):
    """(states[idx], actions[idx], log_probs[idx], values[idx], adv[idx]) in one launch
    (reference src/ppo.py:582-588).  All inputs are flattened-rollout tensors with M rows."""
    lib = _lib.load()
    dev = device()
    ix = idx.to(dev, torch.int64).contiguous()
    st, lp, v, a = (_dev(t, torch.float32) for t in (states, log_probs, values, adv))
    act = actions.to(dev).contiguous()
    assert act.dtype in (torch.int32, torch.float32), "actions must be i32 (Discrete) or f32 (Box)"
    M, B = st.shape[0], ix.numel()
    assert act.shape[0] == M and lp.numel() == M and v.numel() == M and a.numel() == M
    sw = st[0].numel() if M else 0
    aw = act[0].numel() if M else 0
    mb_st = torch.empty((B, *st.shape[1:]), dtype=torch.float32, device=dev)
    mb_act = torch.empty((B, *act.shape[1:]), dtype=act.dtype, device=dev)
    mb_lp, mb_v, mb_a = (torch.empty(B, dtype=torch.float32, device=dev) for _ in range(3))
    err = torch.zeros(1, dtype=torch.int32, device=dev) if check_index else None
    rc = lib.ppo_minibatch_gather(_ptr(ix), B, M, _ptr(st), sw, _ptr(act), aw, _ptr(lp), _ptr(v), _ptr(a),
                                  _ptr(mb_st), _ptr(mb_act), _ptr(mb_lp), _ptr(mb_v), _ptr(mb_a), _ptr(err),
                                  _stream())
    _lib.check(rc, "ppo_minibatch_gather")
    if check_index and int(err.item()):
        raise IndexError("minibatch index out of range")
    return mb_st, mb_act, mb_lp, mb_v, mb_a


# --------------------------------------------------------------------------- update-phase Linear
def linear_wgrad(g: torch.Tensor, x: torch.Tensor, want_bias: bool = True):
    """(dW [M,K], db [M]) of y = x W^T + b from g = dL/dy [B,M] and x [B,K] (ppo_linear_wgrad)."""
    lib = _lib.load()
    g_, x_ = _dev(g, torch.float32), _dev(x, torch.float32)
    B, M = g_.shape
    K = x_.shape[1]
    assert x_.shape[0] == B
    dW = torch.empty((M, K), dtype=torch.float32, device=g_.device)
    db = torch.empty(M, dtype=torch.float32, device=g_.device) if want_bias else None
    ws = _workspace(lib.ppo_linear_wgrad_workspace_bytes(B), g_.device)
    _lib.check(lib.ppo_linear_wgrad(_ptr(g_), _ptr(x_), B, M, K, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), _stream()),
               "ppo_linear_wgrad")
    return dW, db


class _LinearFn(torch.autograd.Function):
    """F.linear whose weight / bias gradients come from ppo_linear_wgrad (one streaming pass over the
    batch) instead of a BLAS GEMM with K = batch size."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        dW, db = linear_wgrad(g, x)
        return gx, dW, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """y = x W^T + b; large 2-D CUDA batches of small layers back-propagate through ppo_linear_wgrad."""
    if (x.is_cuda and x.dim() == 2 and x.shape[0] >= 4096 and weight.shape[0] <= 64 and weight.shape[1] <= 64
            and x.dtype == torch.float32 and bias is not None and torch.is_grad_enabled()
            and (weight.requires_grad or bias.requires_grad)):
        return _LinearFn.apply(x.contiguous(), weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


# --------------------------------------------------------------------------- sampling
def categorical_sample(logits: torch.Tensor, seed: int, env_base: int, step: int, action_out=None, logp_out=None):
    lib = _lib.load()
    lg = _dev(logits, torch.float32)
    N, A = lg.shape
    action = action_out if action_out is not None else torch.empty(N, dtype=torch.int32, device=lg.device)
    logp = logp_out if logp_out is not None else torch.empty(N, dtype=torch.float32, device=lg.device)
    rc = lib.ppo_categorical_sample(_ptr(lg), A, seed, env_base, step, _ptr(action), _ptr(logp), N, _stream())
    _lib.check(rc, "ppo_categorical_sample")
    return action, logp


# --------------------------------------------------------------------------- fused rollout
NET_F32, NET_F16_SPLIT, NET_BF16X3 = 0, 1, 2   # include/ppo_amd.h: PPO_NET_*
NET_AUTO = -1                                  # resolved here, never passed to the library
NET_NAMES = {NET_F32: "f32", NET_F16_SPLIT: "f16_split", NET_BF16X3: "bf16x3", NET_AUTO: "auto"}
# What NET_AUTO runs OUTSIDE the limits of the two-piece f16 split: the f32-grade three-piece bf16 kernels
# (round 3; ~1.7x the speed of the f32-MFMA kernels at the same accuracy).  PPO_AMD_NET_FALLBACK=f32 keeps round 2's.
NET_AUTO_FALLBACK = {"bf16x3": NET_BF16X3, "f32": NET_F32}[os.environ.get("PPO_AMD_NET_FALLBACK", "bf16x3")]
# NET_AUTO: the split kernels run only for parameter sets inside these limits of
# ppo_rollout_pack_params' split_diag (pack.hip says what the three numbers are;
# tests/test_gpu_net_precision.py pins the limits against a float64 forward and PyTorch's f32 one)
SPLIT_FLOOR_MARGIN_MIN = 1e-4   # measured harmless down to 3e-4 (the kernel's own tanh error is larger)
SPLIT_ROWNORM1_MAX = 4.0        # rms row norm of 2 log2(e) W1: 1.0 (CartPole) / 1.25 (Acrobot) at initialisation
SPLIT_ROWNORM2_MAX = 8.5        # ... of 2 log2(e) W2: 4.08 at the reference's orthogonal(sqrt 2) initialisation


def split_ok(diag) -> bool:
    """The AUTO decision from the three host floats of split_diag."""
    return bool(diag[0] >= SPLIT_FLOOR_MARGIN_MIN and diag[1] <= SPLIT_ROWNORM1_MAX and diag[2] <= SPLIT_ROWNORM2_MAX)


_KIND_OF_OBS = {4: 0, 6: 1}


@functools.lru_cache(maxsize=None)
def _rollout_param_floats(kind: int, joint: int, net_mode: int) -> int:
    return int(_lib.load().ppo_rollout_param_floats(kind, joint, net_mode))


def rollout_param_floats(kind: int, joint, net_mode: int) -> int:
    return _rollout_param_floats(int(kind), int(bool(joint)), int(net_mode))


def pack_dense_agent(agent, net_mode: int = NET_F32, out: torch.Tensor | None = None,
                     diag_out: torch.Tensor | None = None) -> torch.Tensor:
    """ppo_rollout_pack_params: the dense Agent's nn.Linear tensors -> the layout k_rollout reads, one
    launch, no host work beyond the call.  net_mode is NET_F32, NET_F16_SPLIT or NET_BF16X3; diag_out (device
    f32[3]) receives split_diag, the accuracy indicators of the split mode for this parameter set."""
    lib = _lib.load()
    assert net_mode in (NET_F32, NET_F16_SPLIT, NET_BF16X3)
    pi = agent.pi_backbone
    w1 = pi[0].weight
    if not w1.is_cuda:
        raise _lib.PpoAmdError("pack_dense_agent needs the agent on the HIP device; there is no CPU fallback")
    kind = _KIND_OF_OBS[w1.shape[1]]
    joint = bool(agent.joint_net)
    tensors = [pi[0].weight, pi[0].bias, pi[2].weight, pi[2].bias]
    if joint:
        tensors += [None] * 4
    else:
        va = agent.va_backbone
        tensors += [va[0].weight, va[0].bias, va[2].weight, va[2].bias]
    tensors += [agent.policy.weight, agent.policy.bias, agent.critic.weight, agent.critic.bias]
    assert agent.policy.weight.shape[0] == (2 if kind == 0 else 3)
    for t in tensors:
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
    n = rollout_param_floats(kind, joint, net_mode)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w1.device)
    assert out.numel() == n and out.dtype == torch.float32 and out.is_cuda
    rc = lib.ppo_rollout_pack_params(kind, int(joint), int(net_mode), *[_ptr(t) for t in tensors], _ptr(out),
                                     _ptr(diag_out), _stream())
    _lib.check(rc, "ppo_rollout_pack_params")
    return out


def resolve_net_mode(agent, net_mode: int):
    """(params, mode actually used, split_ok or None) for a requested mode.  NET_AUTO packs for the
    f16-split kernels, reads split_diag back (one small device->host copy) and falls back to NET_AUTO_FALLBACK
    (the three-piece bf16 layout) when the parameter set is outside that mode's limits."""
    if net_mode != NET_AUTO:
        return pack_dense_agent(agent, net_mode), net_mode, None
    diag = torch.empty(3, dtype=torch.float32, device=agent.policy.weight.device)
    params = pack_dense_agent(agent, NET_F16_SPLIT, diag_out=diag)
    ok = split_ok(diag.tolist())
    if ok:
        return params, NET_F16_SPLIT, True
    return pack_dense_agent(agent, NET_AUTO_FALLBACK), NET_AUTO_FALLBACK, False


def rollout_fused(batch, params, joint, sample_step0, obs_io, states, actions, log_probs, values, rewards, done,
                  trunc, term_obs, ev_dense, T, net_mode: int = NET_F32, reward_sum=None, wave_envs: int = 0, flag_words=None):
    """ppo_rollout_fused over a ppo_amd.envs.DeviceEnvBatch; every tensor is written in place.
    flag_words (gae_flag_words(T, N)): also leave the packed flag bits (ppo_rollout_fused_packed).
    wave_envs != 0 (tests / tuning) forces 32 or 64 envs per wave through the private _ex entry."""
    lib = _lib.load()
    need = rollout_param_floats(batch.kind, int(bool(joint)), int(net_mode))
    assert params.numel() == need and params.dtype == torch.float32 and params.is_cuda, (params.numel(), need)
    bp = batch.pointers()       # (state, elapsed, ep_ret, ep_len, episode, ret_ring, len_ring, ring_cnt): fixed for the batch's life
    args = (
        batch.kind, bp[0], bp[1], bp[2], bp[3], bp[4], None, batch.seed, batch.env_base, params.data_ptr(),
        int(bool(joint)), int(net_mode), int(sample_step0), obs_io.data_ptr(), states.data_ptr(), actions.data_ptr(),
        log_probs.data_ptr(), values.data_ptr(), rewards.data_ptr(), done.data_ptr(), trunc.data_ptr(),
        term_obs.data_ptr(), _ptr(ev_dense))
    tail = (_ptr(reward_sum), bp[5], bp[6], bp[7], batch.ring, batch.num_envs, int(T))
    if flag_words is not None:
        assert flag_words.is_cuda and flag_words.dtype == torch.int32 and flag_words.is_contiguous()
        assert flag_words.shape == ((int(T) + 15) // 16, batch.num_envs)
    if wave_envs:
        rc = lib.ppo_rollout_fused_ex(*args, _ptr(flag_words), *tail, int(wave_envs), _stream())
    elif flag_words is not None:
        rc = lib.ppo_rollout_fused_packed(*args, flag_words.data_ptr(), *tail, _stream())
    else:
        args = args + tail
        rc = lib.ppo_rollout_fused(*args, _stream())
    _lib.check(rc, "ppo_rollout_fused")


# ---- fused dense update (csrc/update.hip) ------------------------------------------------------
_update_ws = {}


class GradGroup(list):
    """[(parameter, gradient)] of one ppo_dense_update_grads launch; the gradients are views of `flat`."""
    flat = None


def dense_update_supported(agent) -> bool:
    """The fused update covers the reference's dense Agent (64-64 tanh) with a Discrete head of 2 or 3
    actions on 4 or 6 observations: CartPole and Acrobot."""
    if agent.conv_net or not agent.discrete:
        return False
    w1 = agent.pi_backbone[0].weight
    return (w1.is_cuda and w1.dtype == torch.float32 and tuple(w1.shape) in ((64, 4), (64, 6))
            and agent.policy.weight.shape[0] in (2, 3))


def dense_update_grads(agent, states, actions, old_log_probs, old_values, advantages, clip_ratio, norm_adv, clip_va_loss,
                       c_pi, c_v, c_ent, stats3=None, idx=None, net_mode: int = NET_F32, out5=None, reuse_bucket=False):
    """ppo_dense_update_grads for a ppo_amd Agent: one launch per backbone.
    Returns (out5, groups): out5 = device f32[5] {policy_loss, value_loss, clip_frac, kl_div, entropy};
    groups = one list of (parameter, gradient tensor) per launch -- [all parameters] for a joint network,
    [policy backbone + policy head, value backbone + critic head] for split networks -- the gradients being
    views of one flat buffer per launch.  Gradient of  c_pi policy_loss + c_v value_loss - c_ent entropy.
    idx (int64 [B], optional): the minibatch is rows idx of the given (flattened rollout) tensors -- the kernel
    gathers as it goes (reference src/ppo.py:582-588); the advantage statistics are those of the minibatch.
    net_mode: NET_F32, or NET_F16_SPLIT for parameter sets inside the split mode's limits (split_ok).
    out5: where to write out5 (device f32[5], e.g. a row of a per-update table) instead of a new tensor.
    reuse_bucket: write the gradients into ONE bucket per (agent, heads) that every such call reuses (the views
    into it are built once; the trainer's loop, where the optimiser step consumes a launch's gradients before the
    next launch is queued) instead of a fresh buffer per call."""
    lib = _lib.load()
    assert net_mode in (NET_F32, NET_F16_SPLIT)
    dev = states.device
    rows = int(states.shape[0])
    B = rows if idx is None else int(idx.numel())
    obs, A = int(states.shape[1]), int(agent.policy.weight.shape[0])
    for t, dt in ((states, torch.float32), (actions, torch.int32), (old_log_probs, torch.float32),
                  (old_values, torch.float32), (advantages, torch.float32)):
        assert t.is_cuda and t.dtype == dt and t.is_contiguous() and t.shape[0] == rows, (t.dtype, t.shape)
    if idx is not None:
        assert idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous() and idx.dim() == 1
    if norm_adv and stats3 is None:
        stats3 = adv_stats(advantages, idx)
    ws = _update_ws.get(dev)
    if ws is None:
        ws = _update_ws[dev] = torch.empty(lib.ppo_dense_update_workspace_bytes(), dtype=torch.uint8, device=dev)
    if out5 is None:       # (every entry is written: heads = 3 by one launch, heads = 1 and 2 by the two of a split network)
        out5 = torch.empty(5, dtype=torch.float32, device=dev)
    assert out5.is_cuda and out5.dtype == torch.float32 and out5.is_contiguous() and out5.numel() == 5
    groups = []
    cache = agent.__dict__.setdefault("_ppo_grad_buckets", {})

    def run(backbone, heads):
        ent = cache.get(heads) if reuse_bucket else None
        l1, l2 = backbone[0], backbone[2]
        if ent is None or ent["w1"] is not l1.weight or ent["dev"] != dev:
            pw, pb = (agent.policy.weight, agent.policy.bias) if heads & 1 else (None, None)
            cw, cb = (agent.critic.weight, agent.critic.bias) if heads & 2 else (None, None)
            params = [l1.weight, l1.bias, l2.weight, l2.bias] + ([pw, pb] if heads & 1 else []) + ([cw, cb] if heads & 2 else [])
            for p in params:
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
            n = lib.ppo_dense_update_grad_floats(obs, A, heads)
            pairs = GradGroup()
            g = pairs.flat = torch.empty(n, dtype=torch.float32, device=dev)
            at = 0
            for p in params:
                pairs.append((p, g[at:at + p.numel()].view_as(p)))
                at += p.numel()
            assert at == n
            ent = {"w1": l1.weight, "dev": dev, "pairs": pairs, "slots": [l1.weight, l1.bias, l2.weight, l2.bias, pw, pb, cw, cb]}
            if reuse_bucket:
                cache[heads] = ent
        pairs = ent["pairs"]
        groups.append(pairs)
        rc = lib.ppo_dense_update_grads(
            obs, A, heads, int(net_mode), *[_ptr(t) for t in ent["slots"]], _ptr(idx), states.data_ptr(), actions.data_ptr(),
            old_log_probs.data_ptr(), old_values.data_ptr(), advantages.data_ptr(), _ptr(stats3), float(clip_ratio),
            int(bool(norm_adv)), int(bool(clip_va_loss)), B, float(c_pi), float(c_v), float(c_ent), out5.data_ptr(),
            pairs.flat.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        _lib.check(rc, "ppo_dense_update_grads")

    if agent.joint_net:
        run(agent.pi_backbone, 3)
    else:
        run(agent.pi_backbone, 1)
        run(agent.va_backbone, 2)
    return out5, groups


# ---- gradient clipping + Adam step in one launch (csrc/optim.hip) ---------------------------------
def clip_adam_supported(opt) -> bool:
    """ppo_clip_adam_step covers the reference's optimiser as it is built (optim.Adam(lr, eps=1e-5),
    src/ppo.py:732-733): plain Adam, one parameter group, no weight decay / amsgrad / maximize."""
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    return not (g.get("amsgrad") or g.get("maximize") or g.get("weight_decay") or g.get("capturable")
                or g.get("fused") or g.get("differentiable")) and not isinstance(g["lr"], torch.Tensor)


_set_version_counter = getattr(getattr(torch._C, "_autograd", None), "_unsafe_set_version_counter", None)   # private torch API
_param_generation = 0


def param_generation() -> int:
    """Number of in-place parameter updates made by this module's kernels (clip_adam_step) so far.
    SyncVecEnv.prepack keys the packed rollout parameters on it as well as on the tensors' own version counters,
    so a torch build without the private version-counter setter still repacks after every optimiser step."""
    return _param_generation


def _note_inplace_update(params):
    """The parameters changed in place behind autograd's back: bump their version counters like an in-place
    torch op would, where this torch build lets us (list form in 2.10, one tensor at a time in older builds)."""
    global _param_generation
    _param_generation += 1
    if _set_version_counter is None:
        return
    try:
        _set_version_counter(params, [p._version + 1 for p in params])
    except TypeError:
        for p in params:
            _set_version_counter(p, p._version + 1)


def clip_adam_step(opt, group: "GradGroup", max_grad_norm: float):
    """nn.utils.clip_grad_norm_(params of `group`, max_grad_norm); opt.step() for the (parameter, gradient)
    pairs of one dense_update_grads launch, as ONE kernel on the group's flat gradient bucket (reference
    src/ppo.py:608-609).  Works on torch.optim.Adam's own state tensors (created here the way Adam's first
    step() creates them), so state_dict / load_state_dict and a later plain opt.step() see the same state."""
    lib = _lib.load()
    params = [p for p, _ in group]
    key = tuple(id(p) for p in params)
    cache = opt.__dict__.setdefault("_ppo_clip_adam", {})
    ent = cache.get(key)
    states = [opt.state[p] for p in params]
    ptrs = tuple(p.data_ptr() for p in params)           # (a parameter whose storage moved -- .to(), .data = -- rebuilds the table)
    if ent is None or ent["pp"] != ptrs or any(st.get("exp_avg") is not m for st, m in zip(states, ent["m"])):
        for p, st in zip(params, states):
            if len(st) == 0:                       # torch/optim/adam.py _init_group
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            for t in (p, st["exp_avg"], st["exp_avg_sq"]):
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            assert st["step"].device.type == "cpu"
        n = len(params)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        # the group's step counters become views of ONE host tensor: one increment per launch instead of n
        # (they stay ordinary 0-d tensors for state_dict and for a plain opt.step())
        steps_all = torch.tensor([float(st["step"]) for st in states], dtype=torch.float32)
        assert bool((steps_all == steps_all[0]).all()), "the parameters of one group have always been stepped together"
        for k, st in enumerate(states):
            st["step"] = steps_all[k]
        ent = cache[key] = {"m": [st["exp_avg"] for st in states], "v": [st["exp_avg_sq"] for st in states],
                            "steps_all": steps_all, "n": n, "pp": ptrs,
                            "p_arr": arr(params), "numel": (C.c_long * n)(*[p.numel() for p in params])}
        ent["m_arr"], ent["v_arr"] = arr(ent["m"]), arr(ent["v"])
    g = opt.param_groups[0]
    ent["steps_all"] += 1
    step = int(ent["steps_all"][0])
    flat = group.flat
    assert flat.is_cuda and flat.dtype == torch.float32 and flat.numel() == sum(p.numel() for p in params)
    rc = lib.ppo_clip_adam_step(ent["n"], ent["p_arr"], ent["m_arr"], ent["v_arr"], ent["numel"], flat.data_ptr(),
                                float(max_grad_norm), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                float(g["eps"]), step, _stream())
    _lib.check(rc, "ppo_clip_adam_step")
    _note_inplace_update(params)
