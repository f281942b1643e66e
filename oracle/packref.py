"""Step-by-step definition of the packed network layout k_rollout reads -- TEST INFRASTRUCTURE ONLY.

The product packs the network with one kernel launch (ppo_rollout_pack_params, csrc/pack.hip); this
file spells the same layout out in plain tensor operations, element by element as documented in
csrc/rollout.hip, so that a GPU test can compare the two bit for bit.  Network being packed:
reference src/ppo.py:124-145 (dense Agent).  Nothing under ppo_amd/ imports this.
"""
import torch

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


NET_F32, NET_F16_SPLIT, NET_BF16X3 = 0, 1, 2      # include/ppo_amd.h: PPO_NET_*
_TANH_SCALE = 2.0 * 1.4426950408889634   # 2 log2(e): tanh(x) = 1 - 2 / (exp2(_TANH_SCALE x) + 1)


def _split_f16(x64: torch.Tensor):
    """x (float64) -> (hi, lo) float16 with x ~= hi + lo (rollout.hip, mode 1); lo may be an f16
    subnormal, which bounds its error by 2^-25 absolute."""
    hi = x64.to(torch.float16)
    lo = (x64 - hi.to(torch.float64)).to(torch.float16)
    return hi, lo


def _bf16_rne(x64: torch.Tensor):
    """float64 -> (bf16 bit pattern as int32, the bf16 value as float64): double -> float32 (to nearest), then
    round to nearest even on the upper 16 bits -- the two roundings pack.hip's bf16_round makes."""
    u = x64.to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFFFFFF
    bits = u >> 16
    val = ((bits << 16).to(torch.int64))
    val = torch.where(val >= 2 ** 31, val - 2 ** 32, val).to(torch.int32).view(torch.float32).to(torch.float64)
    return bits.to(torch.int32), val


def _split_bf16x3(x64: torch.Tensor):
    """x (float64) -> three bf16 bit patterns (int32 in [0, 65536)) with x ~= b0 + b1 + b2 (rollout.hip, mode 2):
    b0 = bf16(x), b1 = bf16(x - b0), b2 = bf16(x - b0 - b1), residuals exact in float64."""
    out = []
    r = x64
    for _ in range(3):
        bits, val = _bf16_rne(r)
        out.append(bits)
        r = r - val
    return out


def _bits16_words(bits: torch.Tensor) -> torch.Tensor:
    """[..., 8] 16-bit patterns (int32) -> [..., 4] float32 holding the same bits (element 2i low, 2i + 1 high)."""
    b = bits.to(torch.int64)
    w = (b[..., 0::2] | (b[..., 1::2] << 16))
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    return w.contiguous().view(torch.float32)


def _f16_words(t16: torch.Tensor) -> torch.Tensor:
    """[..., 8] float16 -> [..., 4] float32 holding the same bits (two f16 per 32-bit word)."""
    return t16.contiguous().view(torch.float32)


def _backbone_f16_split(seq, pieces: int = 2) -> list:
    """One backbone in the mode-1 (pieces = 2: f16) or mode-2 (pieces = 3: bf16) layout: W1[op][piece][lane][4],
    W2[j][op][piece][lane][4], B2[op][r][lane]; weights and biases pre-multiplied by 2 log2(e) in float64
    before the split."""
    def split_words(a):                     # [..., 8] float64 -> list of `pieces` [..., 4] float32 word tensors
        if pieces == 2:
            return [_f16_words(t) for t in _split_f16(a)]
        return [_bits16_words(t) for t in _split_bf16x3(a)]
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
    w1 = torch.stack(split_words(a1), dim=1)                         # [op, piece, lane, 4]
    # layer 2: k-step j = 2 o_in + (r_in >> 3), element i = r_in & 7 -> unit(o_in, r_in, half)
    j = torch.arange(4, device=dev)[:, None, None]
    i = torch.arange(8, device=dev)[None, None, :]
    unit = _d_layout_unit(j >> 1, 8 * (j & 1) + i, half[None, :, None])          # [4, 64, 8]
    a2 = torch.stack([W2[(32 * op + l31)[None, :, None], unit] for op in range(2)], dim=1)   # [j, op, lane, 8]
    w2 = torch.stack(split_words(a2), dim=2)                         # [j, op, piece, lane, 4]
    o = torch.arange(2, device=dev)[:, None, None]
    r = torch.arange(16, device=dev)[None, :, None]
    bb = b2[_d_layout_unit(o, r, half[None, None, :])].to(torch.float32)         # [op, r, lane]
    return [w1.reshape(-1), w2.reshape(-1), bb.reshape(-1)]


def pack_dense_agent_reference(agent, net_mode: int = NET_F32) -> torch.Tensor:
    """Flatten a dense Agent's parameters into the layout k_rollout reads (rollout.hip): per
    backbone the MFMA fragments of both tanh layers -- f32 fragments, the second layer in
    register-chained k order (NET_F32), or two-piece f16 / three-piece bf16 splits of the 2 log2(e)-scaled
    weights (NET_F16_SPLIT / NET_BF16X3) -- then the per-lane head weights HL[4][2][16][64] (j = 0..2 logits zero padded,
    j = 3 critic) and bias4 = (policy bias zero padded to 3, critic bias)."""
    def backbone(seq):
        if net_mode in (NET_F16_SPLIT, NET_BF16X3):
            return _backbone_f16_split(seq, 2 if net_mode == NET_F16_SPLIT else 3)
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
