"""ctypes binding of libppo_amd.so (C ABI declared in include/ppo_amd.h).

There is no fallback: if the shared object is missing or a symbol is absent this module
raises, and every op in ppo_amd.ops raises when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPO_AMD_LIB") or os.path.join(_HERE, "libppo_amd.so")   # (PPO_AMD_LIB: A/B of two builds, development)

ABI_VERSION = 4   # include/ppo_amd.h: PPO_AMD_ABI_VERSION

_vp, _i, _l, _d, _u64, _sz = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/ppo_amd.h one to one
SIGNATURES = {
    "ppo_abi_version": (_i, []),
    "ppo_env_obs_dim": (_i, [_i]),
    "ppo_env_reset": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i, _vp]),
    "ppo_env_step": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ppo_gae_dense": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _vp]),
    "ppo_gae_flag_words": (_sz, [_i, _i]),
    "ppo_gae_dense_packed": (_i, [_vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _vp]),
    "ppo_gae_ragged_workspace_bytes": (_sz, [_i]),
    "ppo_gae_ragged": (_i, [_vp, _vp, _vp, _l, _vp, _vp, _d, _d, _vp, _i, _i, _vp, _sz, _vp, _vp]),
    "ppo_gae_ragged_noalloc": (_i, [_vp, _vp, _vp, _l, _vp, _vp, _d, _d, _vp, _i, _i, _vp, _vp]),
    "ppo_loss_workspace_bytes": (_sz, [_l]),
    "ppo_adv_stats": (_i, [_vp, _l, _vp, _vp, _sz, _vp]),
    "ppo_adv_stats_gather": (_i, [_vp, _vp, _l, _vp, _vp, _sz, _vp]),
    "ppo_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _l, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ppo_loss_logits_fwd_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _l, _vp, _vp, _vp, _vp,
                                     _vp, _sz, _vp]),
    "ppo_dense_update_workspace_bytes": (_sz, []),
    "ppo_dense_update_grad_floats": (_sz, [_i, _i, _i]),
    "ppo_dense_update_grads": (_i, [_i, _i, _i, _i] + [_vp] * 8 + [_vp] * 6 + [_vp, _d, _i, _i, _l, _d, _d, _d, _vp, _vp, _vp, _sz, _vp]),
    "ppo_shuffle_indices": (_i, [_l, _u64, _u64, _vp, _vp]),
    "ppo_adv_stats_epoch_workspace_bytes": (_sz, []),
    "ppo_adv_stats_epoch": (_i, [_vp, _l, _u64, _u64, _l, _vp, _vp, _sz, _vp]),
    "ppo_clip_adam_step": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _d, _l, _vp]),
    "ppo_minibatch_gather": (_i, [_vp, _l, _l, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ppo_linear_wgrad_workspace_bytes": (_sz, [_l]),
    "ppo_linear_wgrad": (_i, [_vp, _vp, _l, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ppo_categorical_sample": (_i, [_vp, _i, _u64, _u64, _u64, _vp, _vp, _i, _vp]),
    "ppo_episode_stats_workspace_bytes": (_sz, []),
    "ppo_episode_stats": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ppo_episode_stats_state_bytes": (_sz, [_i, _i]),
    "ppo_episode_stats_incremental": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ppo_episode_stats_incremental_ints": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ppo_episode_stats_ints_hist_offset": (_sz, []),
    "ppo_episode_stats_hist0_offset": (_sz, []),
    "ppo_episode_hist_level": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "ppo_rollout_param_floats": (_sz, [_i, _i, _i]),
    "ppo_rollout_pack_params": (_i, [_i, _i, _i] + [_vp] * 14 + [_vp]),
    "ppo_rollout_fused": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i, _i, _u64, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ppo_rollout_fused_packed": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i, _i, _u64, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
}
# tuning / test / bench hooks, deliberately not in the public header (listed in INTEGRATION.md)
_PRIVATE = {
    "ppo_gae_dense_variant": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _i, _vp]),
    "ppo_stream_ceiling": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _vp]),
    "ppo_rollout_fused_ex": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i, _i, _u64, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ppo_gae_dense_packed_variant": (_i, [_vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _i, _vp]),
    "ppo_gae_dense_timed": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _i, _i, _i, _vp]),
    "ppo_prof_arm": (_i, [_i, _i]),
    "ppo_prof_next": (_i, []),
    "ppo_test_acrobot_terminal": (_i, [_vp, _vp, _vp, _l, _vp]),
    "ppo_test_acrobot_quot": (_i, [_vp, _vp, _vp, _vp, _l, _vp]),
    "ppo_prof_elapsed_us": (_i, [_i, C.POINTER(C.c_float)]),
}

_lib = None


class PpoAmdError(RuntimeError):
    pass


def load():
    """Load libppo_amd.so once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PpoAmdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ppo_amd/csrc`. ppo_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **_PRIVATE}.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.ppo_abi_version() != ABI_VERSION:
        raise PpoAmdError(f"libppo_amd.so has ABI version {lib.ppo_abi_version()}, this binding expects {ABI_VERSION}: rebuild it "
                          "(make -C ppo_amd/csrc)")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise PpoAmdError(f"{what} failed with hipError_t {rc}"
                          + (" (invalid argument: shape/alignment/NULL check in the C ABI)" if rc == 1 else ""))
