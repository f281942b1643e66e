"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/ppo_amd.h declares.  Only argument validation is exercised -- no kernel is launched."""
import ctypes
import os
import re

import pytest

from ppo_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ppo_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ppo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ppo_amd.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in ppo_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms


def test_every_exported_symbol_is_declared_or_listed_private():
    """nm -D: the library exports nothing beyond include/ppo_amd.h and the private hooks that
    ppo_amd/_lib.py (_PRIVATE) and INTEGRATION.md list."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("ppo_")
                       and ln.split()[-2] in ("T", "t", "W")})
    known = set(declared_symbols()) | set(_lib._PRIVATE)
    assert exported, "nm found no ppo_* symbols"
    assert set(exported) <= known, f"exported but neither declared nor private: {sorted(set(exported) - known)}"
    assert known <= set(exported), f"declared or listed but not exported: {sorted(known - set(exported))}"
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in _lib._PRIVATE:
        assert name in integration, f"private hook {name} is not documented in INTEGRATION.md"


def test_abi_version_and_obs_dim():
    lib = _lib.load()
    assert lib.ppo_abi_version() == _lib.ABI_VERSION == 4
    assert lib.ppo_env_obs_dim(0) == 4 and lib.ppo_env_obs_dim(1) == 6 and lib.ppo_env_obs_dim(7) == -1


def test_header_constants_match_the_python_side():
    from ppo_amd.ppo import _LIB_CONSTS
    text = open(os.path.join(ROOT, "include", "ppo_amd.h")).read()
    for name, value in _LIB_CONSTS.items():
        m = re.search(r"#define\s+%s\s+(\d+)" % name, text)
        assert m and int(m.group(1)) == value, name


def test_argument_validation_rejects_before_launch():
    lib = _lib.load()
    INVALID = 1  # hipErrorInvalidValue
    # unknown env kind, NULL buffers, bad sizes: rejected on the host, nothing is launched
    assert lib.ppo_env_reset(9, None, None, None, None, None, None, 0, 0, None, 4, None) == INVALID
    assert lib.ppo_env_reset(0, None, None, None, None, None, None, 0, 0, None, 4, None) == INVALID
    assert lib.ppo_env_reset(0, None, None, None, None, None, None, 0, 0, None, 0, None) == 0  # empty batch
    assert lib.ppo_gae_dense(None, None, None, None, None, 0.99, 0.95, None, None, 8, 2, None) == INVALID
    assert lib.ppo_gae_ragged(None, None, None, 0, None, None, 0.99, 0.95, None, 8, 2, None, 0, None, None) == INVALID
    assert lib.ppo_loss_fwd_bwd(None, None, None, None, None, None, 0.2, 1, 0, 0, 5, None, None, None, None, 0,
                                None) == INVALID
    assert lib.ppo_adv_stats(None, 5, None, None, 0, None) == INVALID
    assert lib.ppo_categorical_sample(None, 64, 0, 0, 0, None, None, 4, None) == INVALID  # A > 32
    assert lib.ppo_gae_ragged_workspace_bytes(1000) == (1000 + 4) * 8   # counts + one total per 256 columns
    assert lib.ppo_loss_workspace_bytes(10) >= 4 * 8


def test_misaligned_pointers_rejected():
    lib = _lib.load()
    buf = (ctypes.c_char * 4096)()
    base = ctypes.addressof(buf)
    p = base + 4  # 4-byte aligned only
    # state must be 16-byte aligned (double2 loads)
    assert lib.ppo_env_reset(0, p, base, base, base, base, None, 0, 0, base, 1, None) == 1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libppo_amd.so")
    with pytest.raises(_lib.PpoAmdError):
        _lib.load()


def test_ops_refuse_without_hip_device():
    import torch
    from ppo_amd import ops
    if torch.cuda.is_available():
        pytest.skip("HIP device present")
    with pytest.raises(_lib.PpoAmdError):
        ops.gae_dense(*(torch.zeros(2, 2) for _ in range(3)), torch.zeros(2, 2, dtype=torch.int32),
                      torch.zeros(2, 2, dtype=torch.int32), 0.99, 0.95)
