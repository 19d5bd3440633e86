"""tools/isa_audit.py on hand-written ISA snippets: each of the three static checks that gate the hand-scheduled kernels
(tests/test_build_resources.py runs them on the compiled product kernels) must flag its hazard and accept the fixed form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402


def _k(body):
    return ["kern:"] + ["\t" + l if not l.startswith(";;#") and not l.endswith(":") else l for l in body] + ["\ts_endpgm"]


def test_in_flight_register_of_an_inline_asm_load_is_flagged_until_its_counted_wait():
    load = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v0, s[2:3]", ";;#ASMEND"]
    younger = [";;#ASMSTART", "global_load_dwordx4 v[8:11], v0, s[2:3] offset:16", ";;#ASMEND"]
    use = ["v_add_f32_e32 v1, v5, v1"]
    # read before any wait; and with a wait that still leaves the load among the youngest two
    assert isa_audit.audit(_k(load + use))
    assert isa_audit.audit(_k(load + younger + [";;#ASMSTART", "s_waitcnt vmcnt(2)", ";;#ASMEND"] + use))
    # vmcnt(1): only the younger load may be outstanding -> v[4:7] readable, v[8:11] not
    ok = _k(load + younger + [";;#ASMSTART", "s_waitcnt vmcnt(1)", ";;#ASMEND"] + use)
    assert not isa_audit.audit(ok)
    assert isa_audit.audit(_k(load + younger + [";;#ASMSTART", "s_waitcnt vmcnt(1)", ";;#ASMEND", "v_mov_b32_e32 v2, v9"]))
    # both sides of a branch are explored
    br = _k(load + ["s_cbranch_scc1 .LBB0_1", "s_waitcnt vmcnt(0)", ".LBB0_1:"] + use)
    assert isa_audit.audit(br)
    # a scalar load's result still in flight at the end of its statement (the spill the fused kernel hit)
    assert isa_audit.audit(_k([";;#ASMSTART", "s_load_dwordx4 s[8:11], s[2:3], s4", ";;#ASMEND", "v_writelane_b32 v40, s8, 3"]))
    assert not isa_audit.audit(_k([";;#ASMSTART", "s_load_dwordx4 s[8:11], s[2:3], s4", "s_waitcnt lgkmcnt(0)", ";;#ASMEND", "v_writelane_b32 v40, s8, 3"]))


def test_sgpr_written_by_valu_right_in_front_of_an_inline_asm_load_is_flagged():
    reload_ = ["v_readlane_b32 s44, v180, 6", "v_readlane_b32 s45, v180, 7"]
    load = [";;#ASMSTART", "global_load_dwordx4 v[0:3], v9, s[44:45]", ";;#ASMEND"]
    assert isa_audit.sgpr_hazards(_k(reload_ + load))
    assert isa_audit.sgpr_hazards(_k(reload_ + ["v_mov_b32_e32 v1, v2", "s_nop 1"] + load))           # 3 wait states: not enough
    assert not isa_audit.sgpr_hazards(_k(reload_ + ["s_nop 4"] + load))
    assert not isa_audit.sgpr_hazards(_k(reload_ + [";;#ASMSTART", "s_nop 4", "global_load_dwordx4 v[0:3], v9, s[44:45]", ";;#ASMEND"]))
    assert not isa_audit.sgpr_hazards(_k(["s_mov_b32 s44, s10", "s_mov_b32 s45, s11"] + load))        # SALU writes: no hazard
    # through a branch target: the writer sits before the jump
    assert isa_audit.sgpr_hazards(_k(reload_ + ["s_branch .LBB0_2", "v_mov_b32_e32 v1, v2", ".LBB0_2:"] + load))
    # compiler-issued loads are the compiler's business
    assert not isa_audit.sgpr_hazards(_k(reload_ + ["global_load_dwordx4 v[0:3], v9, s[44:45]"]))


def test_packed_fp32_low_lane_reading_the_high_half_of_src1_is_flagged():
    bad = ["v_pk_add_f32 v[0:1], v[0:1], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0] op_sel_hi:[1,0,1]",
           "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]"]
    fine = ["v_pk_add_f32 v[0:1], v[4:5], v[0:1] op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]",
            "v_pk_mov_b32 v[0:1], v[2:3], v[2:3] op_sel:[1,0]", "v_pk_add_f32 v[0:1], v[0:1], v[4:5]"]
    assert len(isa_audit.pk_src1_hi_selects(_k(bad + fine))) == 3
    assert not isa_audit.pk_src1_hi_selects(_k(fine))


def test_loop_exit_flags_keep_the_audit_on_paths_that_exist():
    """hipcc lowers `break` as a flag (s_mov_b64 s[a:b], -1 / 0) tested in a common block: a load requested for the NEXT iteration is
    in flight on the break path too, but that path leaves the loop -- it must not be followed into the loop header."""
    load = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v0, s[2:3]", ";;#ASMEND"]
    body = (load + ["s_cbranch_scc1 .LBB0_BREAK",
                    ";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND", "s_mov_b64 s[16:17], 0", "s_branch .LBB0_JOIN",
                    ".LBB0_BREAK:", "s_mov_b64 s[16:17], -1",
                    ".LBB0_JOIN:", "s_and_b64 vcc, exec, s[16:17]", "s_cbranch_vccnz .LBB0_EXIT",
                    "v_add_f32_e32 v1, v5, v1",              # (the loop header: reads the register)
                    ".LBB0_EXIT:", "s_waitcnt vmcnt(0)"])
    assert not isa_audit.audit(_k(body))
    # without the constant (the flag comes from a comparison) both ways out of the join are possible: flagged
    unknown = [l if l != "s_mov_b64 s[16:17], -1" else "s_cselect_b64 s[16:17], -1, 0" for l in body]
    assert isa_audit.audit(_k(unknown))
