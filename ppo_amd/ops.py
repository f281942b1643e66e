"""Tensor-level wrappers over the C ABI (include/ppo_amd.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; all arithmetic on the
hot path happens in libppo_amd.so.  Tensors that live on the CPU are staged to the HIP
device and the result is staged back -- the computation itself never runs on the CPU, and
with no HIP device every op raises.
"""
from __future__ import annotations

import os

import numpy as np

import torch

from . import _lib

_ws_cache: dict = {}
_GAE_VARIANT = int(os.environ["PPO_GAE_VARIANT"]) if "PPO_GAE_VARIANT" in os.environ else None   # tuning only


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.PpoAmdError("ppo_amd needs a HIP device (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


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


def gae_ragged(rewards, values, end_values, done_flags, trunc_flags, gamma, lam, check_index=True):
    """ppo_gae_ragged: exact drop-in for PPO.compute_advantages (reference src/ppo.py:452-488)."""
    lib = _lib.load()
    src_dev = rewards.device
    r, v, ev = _dev(rewards, torch.float32), _dev(values, torch.float32), _dev(end_values, torch.float32)
    d, tr = _dev(done_flags, torch.int32), _dev(trunc_flags, torch.int32)
    T, N = r.shape
    assert v.shape == (T, N) and d.shape == (T, N) and tr.shape == (T, N) and ev.dim() == 1
    adv = torch.empty((T, N), dtype=torch.float32, device=r.device)
    nws = lib.ppo_gae_ragged_workspace_bytes(N)
    ws = torch.empty(max(nws, 8), dtype=torch.uint8, device=r.device)
    err = torch.zeros(1, dtype=torch.int32, device=r.device)
    rc = lib.ppo_gae_ragged(_ptr(r), _ptr(v), _ptr(ev), ev.numel(), _ptr(d), _ptr(tr), float(gamma), float(lam),
                            _ptr(adv), T, N, _ptr(ws), ws.numel(), _ptr(err), _stream())
    _lib.check(rc, "ppo_gae_ragged")
    if check_index and int(err.item()) != 0:  # torch raises IndexError at src/ppo.py:478
        raise IndexError("index out of range in end_values (done+trunc bookkeeping does not match end_values)")
    return adv if src_dev.type == "cuda" else adv.to(src_dev)


# --------------------------------------------------------------------------- loss
def adv_stats(adv: torch.Tensor) -> torch.Tensor:
    """{count, mean, M2} (fp64, device) of a flat advantage tensor (reference src/ppo.py:501)."""
    lib = _lib.load()
    a = _dev(adv, torch.float32).reshape(-1)
    stats = torch.empty(3, dtype=torch.float64, device=a.device)
    ws = _workspace(lib.ppo_loss_workspace_bytes(a.numel()), a.device)
    _lib.check(lib.ppo_adv_stats(_ptr(a), a.numel(), _ptr(stats), _ptr(ws), ws.numel(), _stream()),
               "ppo_adv_stats")
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
def _d_layout_unit(o, r, half):
    """Hidden unit held by register r of output tile o in lane half `half` of the
    v_mfma_f32_32x32x2_f32 D layout (rollout.hip: unit(o, r, h))."""
    return 32 * o + (r & 3) + 8 * (r >> 2) + 4 * half


def _mfma_frags(weight: torch.Tensor, bias: torch.Tensor, chained: bool = False) -> torch.Tensor:
    """Linear(K -> 64) as v_mfma_f32_32x32x2_f32 A-operand fragments [1 + K/2, 2, 64]
    (layout documented in ppo_amd/csrc/rollout.hip).  chained=True: K = 64 inputs arrive as the
    previous layer's accumulator registers, i.e. k-step 1 + 16 o + r pairs the units
    unit(o, r, 0) and unit(o, r, 1)."""
    W = weight.detach().to(torch.float32)
    K = W.shape[1]
    if K % 2:
        W = torch.cat([W, W.new_zeros(64, 1)], dim=1)
        K += 1
    dev = W.device
    lane = torch.arange(64, device=dev)
    half, l31 = lane >> 5, lane & 31
    s = torch.arange(K // 2, device=dev)
    if chained:
        assert K == 64
        o_in, r_in = s // 16, s % 16
        col = _d_layout_unit(o_in[:, None], r_in[:, None], half[None, :])    # [32 k-steps, 64 lanes]
    else:
        col = 2 * s[:, None] + half[None, :]
    frag = W.new_zeros(1 + K // 2, 2, 64)
    for o in range(2):
        frag[0, o] = torch.where(half == 0, bias.detach().to(torch.float32)[32 * o + l31], W.new_zeros(()))
        frag[1:, o] = W[(32 * o + l31)[None, :], col]
    return frag.reshape(-1)


NET_F32, NET_F16_SPLIT = 0, 1      # include/ppo_amd.h: PPO_NET_F32, PPO_NET_F16_SPLIT
_TANH_SCALE = 2.0 * 1.4426950408889634   # 2 log2(e): tanh(x) = 1 - 2 / (exp2(_TANH_SCALE x) + 1)


def _split_f16(x64: torch.Tensor):
    """x (float64) -> (hi, lo) float16 with x ~= hi + lo (rollout.hip, mode 1); lo may be an f16
    subnormal, which bounds its error by 2^-25 absolute."""
    hi = x64.to(torch.float16)
    lo = (x64 - hi.to(torch.float64)).to(torch.float16)
    return hi, lo


def _f16_words(t16: torch.Tensor) -> torch.Tensor:
    """[..., 8] float16 -> [..., 4] float32 holding the same bits (two f16 per 32-bit word)."""
    return t16.contiguous().view(torch.float32)


def _backbone_f16_split(seq) -> list:
    """One backbone in the mode-1 layout: W1[op][piece][lane][4], W2[j][op][piece][lane][4],
    B2[op][r][lane]; weights and biases pre-multiplied by 2 log2(e) in float64 before the split."""
    W1 = seq[0].weight.detach().to(torch.float64) * _TANH_SCALE      # [64, OBS]
    b1 = seq[0].bias.detach().to(torch.float64) * _TANH_SCALE
    W2 = seq[2].weight.detach().to(torch.float64) * _TANH_SCALE      # [64, 64]
    b2 = seq[2].bias.detach().to(torch.float64) * _TANH_SCALE
    dev = W1.device
    obs = W1.shape[1]
    assert obs < 8
    lane = torch.arange(64, device=dev)
    half, l31 = lane >> 5, lane & 31
    # layer 1: k = i for both lane halves; the bias rides on input slot i = obs (the kernel feeds 1.0 there)
    a1 = W1.new_zeros(2, 64, 8)
    for op in range(2):
        a1[op, :, :obs] = W1[32 * op + l31]
        a1[op, :, obs] = b1[32 * op + l31]
    h1, l1 = _split_f16(a1)
    w1 = torch.stack([_f16_words(h1), _f16_words(l1)], dim=1)        # [op, piece, lane, 4]
    # layer 2: k-step j = 2 o_in + (r_in >> 3), element i = r_in & 7 -> unit(o_in, r_in, half)
    j = torch.arange(4, device=dev)[:, None, None]
    i = torch.arange(8, device=dev)[None, None, :]
    unit = _d_layout_unit(j >> 1, 8 * (j & 1) + i, half[None, :, None])          # [4, 64, 8]
    a2 = torch.stack([W2[(32 * op + l31)[None, :, None], unit] for op in range(2)], dim=1)   # [j, op, lane, 8]
    h2, l2 = _split_f16(a2)
    w2 = torch.stack([_f16_words(h2), _f16_words(l2)], dim=2)        # [j, op, piece, lane, 4]
    o = torch.arange(2, device=dev)[:, None, None]
    r = torch.arange(16, device=dev)[None, :, None]
    bb = b2[_d_layout_unit(o, r, half[None, None, :])].to(torch.float32)         # [op, r, lane]
    return [w1.reshape(-1), w2.reshape(-1), bb.reshape(-1)]


def pack_dense_agent_reference(agent, net_mode: int = NET_F32) -> torch.Tensor:
    """Straightforward form of pack_dense_agent (dozens of small tensor ops; the definition the
    planned form below is tested against).  Flatten a dense Agent's parameters into the layout k_rollout reads (rollout.hip): per
    backbone the MFMA fragments of both tanh layers -- f32 fragments, the second layer in
    register-chained k order (NET_F32), or two-piece f16 splits of the 2 log2(e)-scaled weights
    (NET_F16_SPLIT) -- then the per-lane head weights HL[4][2][16][64] (j = 0..2 logits zero padded,
    j = 3 critic) and bias4 = (policy bias zero padded to 3, critic bias)."""
    def backbone(seq):
        if net_mode == NET_F16_SPLIT:
            return _backbone_f16_split(seq)
        return [_mfma_frags(seq[0].weight, seq[0].bias), _mfma_frags(seq[2].weight, seq[2].bias, chained=True)]
    with torch.no_grad():
        parts = backbone(agent.pi_backbone)
        if not agent.joint_net:
            parts += backbone(agent.va_backbone)
        A = agent.policy.weight.shape[0]
        assert A <= 3
        dev = agent.policy.weight.device
        wh = agent.policy.weight.new_zeros(4, 64)
        wh[:A] = agent.policy.weight
        wh[3] = agent.critic.weight.reshape(-1)
        lane = torch.arange(64, device=dev)
        o = torch.arange(2, device=dev)[:, None, None]
        r = torch.arange(16, device=dev)[None, :, None]
        unit = _d_layout_unit(o, r, (lane >> 5)[None, None, :])              # [2, 16, 64]
        hl = wh[:, unit]                                                     # [4, 2, 16, 64]
        bias4 = agent.policy.bias.new_zeros(4)
        bias4[:A] = agent.policy.bias
        bias4[3] = agent.critic.bias[0]
        parts += [hl.reshape(-1), bias4]
        return torch.cat([p.detach().to(torch.float32).reshape(-1) for p in parts]).contiguous()


class _PackPlan:
    """pack_dense_agent as a handful of gathers.  The layout never changes for a given network shape,
    so it is worked out once -- by running the reference packing code on tensors of INDICES into the
    flattened parameter vector instead of on values -- and a pack is then: concatenate the
    parameters, gather, scale, split into f16 pieces, gather into place.  (The step-by-step form
    costs 0.6-1.0 ms of small launches per rollout in training, as much as the rollout kernel.)"""

    def __init__(self, agent, net_mode):
        dev = agent.policy.weight.device
        self.names = [n for n, _ in agent.named_parameters()]
        shapes = {n: p.shape for n, p in agent.named_parameters()}
        offs, total = {}, 0
        for n in self.names:
            offs[n] = total
            total += int(np.prod(shapes[n]))
        self.pad = total                                   # index of the appended 0.0

        def idx(name):
            return (offs[name] + torch.arange(int(np.prod(shapes[name])), device=dev)).reshape(shapes[name])

        def seq_idx(prefix):                               # Sequential(Linear, Tanh, Linear, Tanh)
            return idx(prefix + ".0.weight"), idx(prefix + ".0.bias"), idx(prefix + ".2.weight"), idx(prefix + ".2.bias")

        lane = torch.arange(64, device=dev)
        half, l31 = lane >> 5, lane & 31
        pad = self.pad
        seg16, seg32 = [], []                              # (src, is_lo) element lists; (src, scale) lists
        order = []                                         # final layout: ("h", n_words) / ("f", n_floats) in sequence
        backbones = ["pi_backbone"] + ([] if agent.joint_net else ["va_backbone"])
        for bbn in backbones:
            W1, b1, W2, b2 = seq_idx(bbn)
            obs = W1.shape[1]
            if net_mode == NET_F16_SPLIT:
                assert obs < 8
                a1 = torch.full((2, 64, 8), pad, dtype=torch.long, device=dev)
                for op in range(2):
                    a1[op, :, :obs] = W1[32 * op + l31]
                    a1[op, :, obs] = b1[32 * op + l31]
                j = torch.arange(4, device=dev)[:, None, None]
                i = torch.arange(8, device=dev)[None, None, :]
                unit = _d_layout_unit(j >> 1, 8 * (j & 1) + i, half[None, :, None])
                a2 = torch.stack([W2[(32 * op + l31)[None, :, None], unit] for op in range(2)], dim=1)   # [j, op, lane, 8]
                for src, piece_dim in ((a1, 1), (a2, 2)):
                    e = torch.stack([src, src], dim=piece_dim)                  # hi block, lo block
                    lo = torch.zeros_like(e, dtype=torch.bool)
                    lo.select(piece_dim, 1).fill_(True)
                    seg16.append((e.reshape(-1), lo.reshape(-1)))
                    order.append(("h", e.numel() // 2))
                o = torch.arange(2, device=dev)[:, None, None]
                r = torch.arange(16, device=dev)[None, :, None]
                bsrc = b2[_d_layout_unit(o, r, half[None, None, :])].reshape(-1)
                seg32.append((bsrc, torch.full_like(bsrc, 0, dtype=torch.float64) + _TANH_SCALE))
                order.append(("f", bsrc.numel()))
            else:
                for W, b, chained in ((W1, b1, False), (W2, b2, True)):
                    K = W.shape[1]
                    if K % 2:
                        W = torch.cat([W, torch.full((64, 1), pad, dtype=torch.long, device=dev)], dim=1)
                        K += 1
                    sidx = torch.arange(K // 2, device=dev)
                    if chained:
                        col = _d_layout_unit((sidx // 16)[:, None], (sidx % 16)[:, None], half[None, :])
                    else:
                        col = 2 * sidx[:, None] + half[None, :]
                    frag = torch.full((1 + K // 2, 2, 64), pad, dtype=torch.long, device=dev)
                    for o in range(2):
                        frag[0, o] = torch.where(half == 0, b[32 * o + l31], torch.full_like(l31, pad))
                        frag[1:, o] = W[(32 * o + l31)[None, :], col]
                    seg32.append((frag.reshape(-1), torch.ones(frag.numel(), dtype=torch.float64, device=dev)))
                    order.append(("f", frag.numel()))
        A = shapes["policy.weight"][0]
        assert A <= 3
        wh = torch.full((4, 64), pad, dtype=torch.long, device=dev)
        wh[:A] = idx("policy.weight")
        wh[3] = idx("critic.weight").reshape(-1)
        o = torch.arange(2, device=dev)[:, None, None]
        r = torch.arange(16, device=dev)[None, :, None]
        hl = wh[:, _d_layout_unit(o, r, (lane >> 5)[None, None, :])].reshape(-1)
        bias4 = torch.full((4,), pad, dtype=torch.long, device=dev)
        bias4[:A] = idx("policy.bias")
        bias4[3] = idx("critic.bias")[0]
        tail = torch.cat([hl, bias4])
        seg32.append((tail, torch.ones(tail.numel(), dtype=torch.float64, device=dev)))
        order.append(("f", tail.numel()))
        # element lists, and where each output word comes from in cat([f16 words, f32 values])
        self.src16 = torch.cat([e for e, _ in seg16]) if seg16 else None
        self.lo16 = torch.cat([m for _, m in seg16]) if seg16 else None
        self.src32 = torch.cat([e for e, _ in seg32])
        self.scale32 = torch.cat([c for _, c in seg32])
        n_words = 0 if self.src16 is None else self.src16.numel() // 2
        perm, at_h, at_f = [], 0, n_words
        for kind, n in order:
            if kind == "h":
                perm.append(torch.arange(at_h, at_h + n, device=dev)); at_h += n
            else:
                perm.append(torch.arange(at_f, at_f + n, device=dev)); at_f += n
        self.perm = torch.cat(perm)

    def pack(self, agent) -> torch.Tensor:
        params = dict(agent.named_parameters())
        flat = [params[n].detach().reshape(-1) for n in self.names]
        theta = torch.cat(flat + [flat[0].new_zeros(1)]).to(torch.float64)
        parts = []
        if self.src16 is not None:
            v = theta[self.src16] * _TANH_SCALE
            hi = v.to(torch.float16)
            lo = (v - hi.to(torch.float64)).to(torch.float16)
            parts.append(torch.where(self.lo16, lo, hi).view(torch.float32))
        parts.append((theta[self.src32] * self.scale32).to(torch.float32))
        return torch.cat(parts)[self.perm].contiguous()


_PACK_PLANS: dict = {}


def pack_dense_agent(agent, net_mode: int = NET_F32) -> torch.Tensor:
    """Flatten a dense Agent's parameters into the layout k_rollout reads (rollout.hip; element by
    element the same as pack_dense_agent_reference).  The gather plan is cached per network shape,
    mode and device."""
    key = (int(net_mode), bool(agent.joint_net), str(agent.policy.weight.device),
           tuple((n, tuple(p.shape)) for n, p in agent.named_parameters()))
    plan = _PACK_PLANS.get(key)
    if plan is None:
        plan = _PACK_PLANS[key] = _PackPlan(agent, net_mode)
    with torch.no_grad():
        return plan.pack(agent)


def rollout_fused(batch, params, joint, sample_step0, obs_io, states, actions, log_probs, values, rewards, done,
                  trunc, term_obs, ev_dense, T, net_mode: int = NET_F32, reward_sum=None, wave_envs: int = 0):
    """ppo_rollout_fused over a ppo_amd.envs.DeviceEnvBatch; every tensor is written in place.
    wave_envs != 0 (tests / tuning) forces 32 or 64 envs per wave through the private _ex entry."""
    lib = _lib.load()
    need = lib.ppo_rollout_param_floats(batch.kind, int(bool(joint)), int(net_mode))
    assert params.numel() == need and params.dtype == torch.float32 and params.is_cuda, (params.numel(), need)
    args = (
        batch.kind, batch.state.data_ptr(), batch.elapsed.data_ptr(), batch.ep_ret.data_ptr(),
        batch.ep_len.data_ptr(), batch.episode.data_ptr(), None, batch.seed, batch.env_base, params.data_ptr(),
        int(bool(joint)), int(net_mode), int(sample_step0), obs_io.data_ptr(), states.data_ptr(), actions.data_ptr(),
        log_probs.data_ptr(), values.data_ptr(), rewards.data_ptr(), done.data_ptr(), trunc.data_ptr(),
        term_obs.data_ptr(), _ptr(ev_dense), _ptr(reward_sum), batch.ret_ring.data_ptr(), batch.len_ring.data_ptr(), batch.ring_cnt.data_ptr(),
        batch.ring, batch.num_envs, int(T))
    if wave_envs:
        rc = lib.ppo_rollout_fused_ex(*args, int(wave_envs), _stream())
    else:
        rc = lib.ppo_rollout_fused(*args, _stream())
    _lib.check(rc, "ppo_rollout_fused")
