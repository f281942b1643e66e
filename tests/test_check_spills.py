"""CPU: tools/check_spills.py, the build-time check of hipcc's device assembly (ppo_amd/csrc/Makefile runs it on every
object).  The first listing is the shape of what hipcc 7.2 generated for one rollout kernel in round 3:
a 128-bit weight fragment spilled as three dwords to scratch and one to an AGPR, reloaded as three dwords only."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_spills", os.path.join(ROOT, "tools", "check_spills.py"))
check_spills = importlib.util.module_from_spec(spec)
spec.loader.exec_module(check_spills)

LOST = """
_Z6kernelPf:                            ; @_Z6kernelPf
\tglobal_load_dwordx4 v[10:13], v[6:7], off offset:1024
\ts_waitcnt vmcnt(0)
\tv_accvgpr_write_b32 a239, v13           ;  Reload Reuse
\tscratch_store_dwordx3 off, v[10:12], off offset:8 ; 12-byte Folded Spill
\tglobal_load_dwordx4 v[226:229], v[6:7], off
\tscratch_load_dwordx3 a[220:222], off, off offset:8 ; 12-byte Folded Reload
\ts_waitcnt vmcnt(0)
\tv_mfma_f32_32x32x16_bf16 v[66:81], a[220:223], v[226:229], 0
\tglobal_store_dwordx4 v[6:7], v[66:69], off
\ts_endpgm
"""

WHOLE = LOST.replace("\tscratch_load_dwordx3 a[220:222], off, off offset:8 ; 12-byte Folded Reload\n",
                     "\tscratch_load_dwordx3 a[220:222], off, off offset:8 ; 12-byte Folded Reload\n"
                     "\tv_accvgpr_mov_b32 a223, a239\n")


def _run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return check_spills.check_file(str(p))


def test_lost_part_of_a_split_spill_is_reported(tmp_path):
    kernels, findings = _run(tmp_path, LOST)
    assert kernels == 1
    assert any("reads a223" in f for f in findings)              # the matrix-core instruction's operand nothing wrote
    assert any("writes a239" in f and "nothing in the kernel reads it" in f for f in findings)


def test_complete_reload_passes(tmp_path):
    kernels, findings = _run(tmp_path, WHOLE)
    assert kernels == 1 and findings == []


def test_a_call_inside_a_kernel_is_reported(tmp_path):
    """hipcc outlined a lambda of the rollout kernel once (three call sites): the callee returned wrong values"""
    text = WHOLE.replace("\ts_waitcnt vmcnt(0)\n\tv_mfma", "\ts_getpc_b64 s[0:1]\n\ts_swappc_b64 s[30:31], s[0:1]\n\ts_waitcnt vmcnt(0)\n\tv_mfma")
    assert "s_swappc" in text
    kernels, findings = _run(tmp_path, text)
    assert kernels == 1 and len(findings) == 1 and "function call inside the kernel" in findings[0]


def test_build_runs_the_check():
    mk = open(os.path.join(ROOT, "ppo_amd", "csrc", "Makefile")).read()
    assert "check_spills.py" in mk and "-save-temps=obj" in mk and ".DELETE_ON_ERROR" in mk
