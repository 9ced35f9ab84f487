"""tools/isa_lint.py: the build-time check for wait-state hazards around inline asm (CPU only).

The two R1 patterns are the ones round 2's review quoted from the shipped code objects: a store base
restored from a spill lane by v_readlane directly in front of an asm store that reads it.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def lint_text(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("\t.text\n\t.globl\tkern\n\t.type\tkern,@function\nkern:\n" + textwrap.dedent(body) +
                 "\ts_endpgm\n.Lfunc_end0:\n")
    return isa_lint.lint_file(str(p))


def rules(found):
    return sorted(f[3] for f in found)


def test_readlane_restore_in_front_of_asm_store_x4(tmp_path):
    # sync_split_kernel<c128, gardner2, true, NWV=4, H=4>, round 2
    f = lint_text(tmp_path, """\
        \tv_lshl_add_u32 v4, s45, 10, v20
        \tv_readlane_b32 s1, v127, 37
        \t;;#ASMSTART
        \tglobal_store_dwordx4 v4, v[0:3], s[0:1]
        \ts_nop 1
        \t;;#ASMEND
    """)
    assert rules(f) == ["R1"] and "s1 0 wait" in f[0][4] and "inline asm" in f[0][4]


def test_readlane_restore_in_front_of_asm_store_x2(tmp_path):
    # the stamps build of the default kernel, round 2
    f = lint_text(tmp_path, """\
        \tv_readlane_b32 s5, v108, 25
        \t;;#ASMSTART
        \tglobal_store_dwordx2 v6, v[4:5], s[4:5]
        \t;;#ASMEND
    """)
    assert rules(f) == ["R1"]


@pytest.mark.parametrize("pad,clean", [(0, False), (3, False), (4, True)])
def test_s_nop_counts_k_plus_one(tmp_path, pad, clean):
    f = lint_text(tmp_path, f"""\
        \tv_readfirstlane_b32 s4, v1
        \ts_nop {pad}
        \tglobal_store_dword v6, v4, s[4:5]
    """)
    assert (not f) == clean


def test_salu_copy_inside_the_asm_is_safe(tmp_path):
    f = lint_text(tmp_path, """\
        \tv_readlane_b32 s7, v127, 37
        \t;;#ASMSTART
        \ts_mov_b64 s[0:1], s[6:7]
        \tglobal_store_dwordx4 v4, v[0:3], s[0:1]
        \ts_nop 1
        \t;;#ASMEND
        \tv_mov_b32 v0, 0
    """)
    assert not f


def test_compare_result_as_store_base_and_carry_out(tmp_path):
    f = lint_text(tmp_path, """\
        \tv_cmp_lt_f64_e64 s[8:9], v[0:1], v[2:3]
        \tv_add_co_u32_e64 v1, s[10:11], v2, v3
        \tglobal_load_dword v5, v6, s[10:11]
        \tglobal_load_dword v5, v6, s[8:9]
    """)
    assert rules(f) == ["R1"] * 4


def test_hazard_through_a_branch(tmp_path):
    # the restore sits at the end of one predecessor, the store at the head of the join block
    f = lint_text(tmp_path, """\
        \ts_cbranch_scc1 .LBB0_2
        \tv_readlane_b32 s3, v9, 1
        \ts_branch .LBB0_3
        .LBB0_2:
        \ts_nop 7
        \ts_nop 7
        .LBB0_3:
        \tglobal_store_dword v0, v1, s[2:3]
    """)
    assert rules(f) == ["R1"]


def test_wide_store_data_overwritten_too_early(tmp_path):
    bad = lint_text(tmp_path, """\
        \t;;#ASMSTART
        \tglobal_store_dwordx4 v4, v[0:3], s[0:1]
        \t;;#ASMEND
        \tv_mov_b32 v2, 0
    """)
    assert rules(bad) == ["R2"]
    good = lint_text(tmp_path, """\
        \t;;#ASMSTART
        \tglobal_store_dwordx4 v4, v[0:3], s[0:1]
        \ts_nop 1
        \t;;#ASMEND
        \tv_mov_b32 v2, 0
    """)
    assert not good


def test_m0_in_front_of_lds_dma(tmp_path):
    bad = lint_text(tmp_path, "\ts_mov_b32 m0, s4\n\tglobal_load_lds_dwordx4 v[0:1], off\n")
    assert rules(bad) == ["R3"]
    good = lint_text(tmp_path, "\ts_mov_b32 m0, s4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[0:1], off\n")
    assert not good
    valu = lint_text(tmp_path, "\tv_readfirstlane_b32 m0, v3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[0:1], off\n")
    assert rules(valu) == ["R1"]


def test_cli_exit_status(tmp_path):
    p = tmp_path / "bad.s"
    p.write_text("kern:\n\tv_readlane_b32 s1, v127, 37\n\tglobal_store_dword v4, v0, s[0:1]\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "R1" in r.stdout


def test_the_build_ran_the_lint():
    """`make` links libdsp_hip.so only after the lint passed over every object's assembly."""
    mk = open(os.path.join(ROOT, "dsp_amd", "csrc", "Makefile")).read()
    assert "$(OUT): $(OBJS) lint.ok" in mk
    for flavour in ("trace: $(TOBJS) lint_trace.ok", "stamps: $(SOBJS) lint_stamps.ok"):
        assert flavour in mk
