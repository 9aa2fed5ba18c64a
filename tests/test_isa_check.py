"""rag_dpo_amd/isa_check.py on hand-written listings: the three properties the build demands of the scan kernels' ISA
(DESIGN.md §10 "the RDX_CHECK_BOUNDS fault"), each with a listing that breaks it and one that keeps it."""
from rag_dpo_amd import isa_check

HEAD = "_ZN3rdx6k_scanILi64ELi1EEEvNS_10ScanParamsE:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"


def asm(*lines):
    return "".join(f"\t;;#ASMSTART\n\t{l}\n\t;;#ASMEND\n" for l in lines)


def hazards(body):
    res = isa_check.check_listing(HEAD + body + TAIL)
    assert len(res) == 1
    return [h["kind"] for h in next(iter(res.values()))[0]]


LOOP = (asm("global_load_dwordx4 v[2:5], v90, s[4:5] offset:0") +
        asm("global_load_dwordx4 v[6:9], v90, s[4:5] offset:0x400") +
        ".LBB0_1:\n" +
        asm("s_waitcnt vmcnt(1)") +                                   # retires the load into v[2:5], keeps the younger one
        "\tv_mfma_f32_16x16x32_f16 v[20:23], v[2:5], v[30:33], v[20:23]\n" +
        asm("global_load_dwordx4 v[2:5], v90, s[4:5] offset:0") +     # refill behind the last reader
        asm("s_waitcnt vmcnt(1)") +
        "\tv_mfma_f32_16x16x32_f16 v[24:27], v[6:9], v[30:33], v[24:27]\n" +
        asm("global_load_dwordx4 v[6:9], v90, s[4:5] offset:0x400") +
        "\ts_cmp_lg_u32 s6, 0\n\ts_cbranch_scc1 .LBB0_1\n" +
        asm("s_waitcnt vmcnt(0)"))


def test_clean_pipeline_passes():
    assert hazards(LOOP) == []


def test_compiler_copy_of_an_in_flight_fragment_is_refused():
    body = LOOP.replace("\tv_mfma_f32_16x16x32_f16 v[24:27]", "\tv_mov_b32_e32 v40, v3\n\tv_mfma_f32_16x16x32_f16 v[24:27]")
    hz = hazards(body)                     # v3 was refilled one instruction earlier and is in flight
    assert any("foreign instruction" in h for h in hz)
    body = LOOP.replace("\ts_cmp_lg_u32 s6, 0", "\tscratch_store_dwordx4 off, v[6:9], s33\n\ts_cmp_lg_u32 s6, 0")   # a spill
    assert any("foreign instruction" in h for h in hazards(body))
    body = LOOP.replace("v_mfma_f32_16x16x32_f16 v[24:27], v[6:9]", "v_mfma_f32_16x16x32_f16 v[2:5], v[6:9]")       # an MFMA WRITING it
    assert any("foreign instruction" in h for h in hazards(body))


def test_wrong_wait_count_is_refused():
    body = LOOP.replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)")        # leaves the consumed fragment in flight on every path
    assert any("before its load can have retired" in h for h in hazards(body))


def test_rare_path_traffic_does_not_hide_or_fake_a_hazard():
    # a store on a rarely taken path (the emit path) adds a younger operation on THAT path only: the consumers stay legal, and a
    # foreign touch on the common path is still seen
    rare = ("\ts_cbranch_scc0 .LBB0_2\n\tglobal_store_dwordx2 v50, v[51:52], s[8:9]\n.LBB0_2:\n")
    body = LOOP.replace("\ts_cmp_lg_u32 s6, 0\n", rare + "\ts_cmp_lg_u32 s6, 0\n")
    assert hazards(body) == []
    assert hazards(body.replace(".LBB0_2:\n", ".LBB0_2:\n\tv_add_u32_e32 v7, v7, v7\n"))


def test_valu_written_sgpr_needs_five_wait_states_before_an_asm_load():
    pre = "\tv_readfirstlane_b32 s15, v1\n\tv_readfirstlane_b32 s14, v0\n"
    ld = asm("global_load_dwordx4 v[2:5], v90, s[14:15] offset:0") + asm("s_waitcnt vmcnt(0)")
    assert any("wait state" in h for h in hazards(pre + ld))                        # what the branchy RDX_CHECK_BOUNDS build contained
    assert any("wait state" in h for h in hazards(pre + "\ts_nop 2\n" + ld))        # 3 + 1 wait states: still short for s14
    assert hazards(pre + "\ts_nop 4\n" + ld) == []
    assert hazards(pre + "\ts_add_u32 s20, s20, 1\n" * 5 + ld) == []
    salu = "\ts_add_u32 s14, s20, s22\n\ts_addc_u32 s15, s21, s23\n"                 # SALU-written address: no such hazard
    assert hazards(salu + ld) == []


def test_the_shipped_library_passed():
    import json
    from rag_dpo_amd import build
    build.build_lib()
    rec = json.load(open(build.RESOURCES))["_isa_check"]
    assert rec["hazards"] == 0 and rec["kernels"] >= 16 and rec["asm_loads"] >= 16 * 12 and len(rec["checks"]) == 3
