"""tools/check_async_loads.py is a correctness gate of the build (the pipelined march kernels issue their gathers as inline
asm and retire them with hand-counted waits): its own regexes and data-flow are pinned here on canned listings, so a change of
llvm-objdump's format or of the script cannot turn the gate into a silent pass."""
import importlib.util
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("check_async_loads", ROOT / "tools" / "check_async_loads.py")
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def listing(rows):
    """rows of (instruction text, size in bytes) -> [(line number, objdump-style line)] at consecutive addresses"""
    out, addr = [], 0x1000
    for no, (text, size) in enumerate(rows):
        words = " ".join(["DEADBEEF"] * (size // 4))
        out.append((no, f"\t{text:<60} // {addr:012X}: {words}"))
        addr += size
    return out


GOOD = [("global_load_dwordx4 v[0:3], v20, s[4:5]", 8), ("global_load_dwordx4 v[4:7], v21, s[4:5]", 8),
        ("v_add_u32_e32 v22, v20, v21", 4),                       # touches neither destination
        ("s_waitcnt vmcnt(1)", 4), ("v_add_f32_e32 v30, v0, v1", 4),   # the older gather has landed
        ("s_waitcnt vmcnt(0)", 4), ("v_add_f32_e32 v31, v4, v5", 4), ("s_endpgm", 4)]


def test_a_clean_listing_passes():
    assert chk.check("k", listing(GOOD)) == []


def test_a_read_before_the_wait_is_reported():
    bad = list(GOOD)
    bad[4], bad[3] = bad[3], bad[4]                               # the use of v0 moves above `vmcnt(1)`
    found = chk.check("k", listing(bad))
    assert len(found) == 1 and "v_add_f32_e32 v30, v0, v1" in found[0][1]


def test_a_use_of_the_newer_gather_after_a_partial_wait_is_reported():
    rows = GOOD[:4] + [("v_mov_b32_e32 v40, v6", 4)] + GOOD[4:]    # v[4:7] is still in flight after vmcnt(1)
    found = chk.check("k", listing(rows))
    assert [t for _, t in found] == ["v_mov_b32_e32 v40, v6"]


def test_an_overwrite_and_an_address_use_of_an_inflight_destination_are_reported():
    rows = [("global_load_dwordx2 v[8:9], v20, s[4:5]", 8), ("v_mov_b32_e32 v9, 0", 4),           # write
            ("global_load_dword v10, v8, s[4:5]", 8),                                             # address register = a destination
            ("s_waitcnt vmcnt(0)", 4), ("s_endpgm", 4)]
    found = chk.check("k", listing(rows))
    assert [t for _, t in found] == ["v_mov_b32_e32 v9, 0", "global_load_dword v10, v8, s[4:5]"]


def test_a_violation_that_only_exists_around_the_loop_back_edge_is_reported():
    # the gather at the bottom of the loop body is still in flight when the back edge reaches the read at the top
    rows = [("s_waitcnt vmcnt(0)", 4),
            ("v_add_f32_e32 v30, v0, v1", 4),                     # <- loop header: reads v[0:3]
            ("global_load_dwordx4 v[0:3], v20, s[4:5]", 8),
            ("s_cbranch_vccnz -4", 4),                            # back to the header: 4 bytes after this + (-4) * 4 = the v_add
            ("s_waitcnt vmcnt(0)", 4), ("s_endpgm", 4)]
    lst = listing(rows)
    # target address = address after the branch + 4 * simm16; header is 3 instructions (16 bytes) back
    found = chk.check("k", lst)
    # (the re-issue into registers whose previous gather never got a wait is a second violation of the same loop)
    assert [t for _, t in found] == ["v_add_f32_e32 v30, v0, v1", "global_load_dwordx4 v[0:3], v20, s[4:5]"]


def test_the_tool_directory_comes_from_the_environment(monkeypatch, tmp_path):
    for t in chk.TOOLS:
        (tmp_path / t).write_text("")
    monkeypatch.setenv("MRIRT_LLVM_BIN", str(tmp_path))
    assert chk.llvm_bin() == tmp_path


# ---- the VALU-writes-SGPR -> vector-memory-reads-it hazard (5 wait states on gfx9, not interlocked) ----------------------------
def test_a_base_pointer_restored_by_readlane_right_before_a_gather_is_reported():
    """The root cause of the memory-access fault of rounds 3 / 4, as it stood in brats_march_pipe_kernel<strict, QUAD, 3, labels>."""
    rows = [("v_readlane_b32 s60, v110, 6", 8), ("v_trunc_f32_e32 v48, v46", 4), ("v_med3_f32 v64, v64, s24, 0", 8),
            ("v_readlane_b32 s61, v110, 7", 8), ("v_sub_f32_e32 v49, v46, v48", 4), ("v_trunc_f32_e32 v90, v64", 4),
            ("global_load_dwordx4 v[26:29], v42, s[60:61]", 8), ("s_waitcnt vmcnt(0)", 4), ("s_endpgm", 4)]
    found = chk.hazard_scan("k", listing(rows))
    assert len(found) == 1 and found[0][2] == "v_readlane_b32 s61, v110, 7" and found[0][3] == 2


def test_five_wait_states_or_a_scalar_copy_clear_the_hazard():
    padded = [("v_readlane_b32 s61, v110, 7", 8), ("s_nop 4", 4), ("global_load_dwordx4 v[26:29], v42, s[60:61]", 8), ("s_endpgm", 4)]
    assert chk.hazard_scan("k", listing(padded)) == []
    short = [("v_readlane_b32 s61, v110, 7", 8), ("s_nop 2", 4), ("global_load_dwordx4 v[26:29], v42, s[60:61]", 8), ("s_endpgm", 4)]
    assert len(chk.hazard_scan("k", listing(short))) == 1
    # what the asm gathers do now: the vector-memory instruction reads an SGPR pair a SCALAR move wrote
    copied = [("v_readlane_b32 s61, v110, 7", 8), ("s_mov_b64 s[30:31], s[60:61]", 4), ("global_load_dwordx4 v[26:29], v42, s[30:31]", 8), ("s_endpgm", 4)]
    assert chk.hazard_scan("k", listing(copied)) == []


def test_the_hazard_is_found_across_a_branch_and_for_compare_and_carry_writers():
    # the writer sits in a predecessor block that jumps to the load
    rows = [("v_cmp_gt_f32_e64 s[8:9], v1, v2", 8), ("s_branch 1", 4), ("s_nop 7", 4),
            ("buffer_load_dword v3, v4, s[8:11], 0 offen", 8), ("s_endpgm", 4)]
    found = chk.hazard_scan("k", listing(rows))
    assert len(found) == 1 and found[0][2].startswith("v_cmp_gt_f32_e64")
    carry = [("v_div_scale_f32 v7, s[34:35], s30, s30, v6", 8), ("global_load_dwordx4 v[8:11], v5, s[34:35]", 8), ("s_endpgm", 4)]
    assert len(chk.hazard_scan("k", listing(carry))) == 1
    vcc = [("v_cmp_eq_u32_e32 vcc, 1, v7", 4), ("global_load_dword v1, v2, vcc", 8), ("s_endpgm", 4)]
    assert len(chk.hazard_scan("k", listing(vcc))) == 1


def test_a_scalar_redefinition_between_the_valu_write_and_the_load_clears_the_register():
    """What the fixed gathers look like when the allocator reuses a mask register for the copied base: the v_cmp's result is dead,
    the s_mov_b64 defines the pair the load reads."""
    rows = [("v_cmp_gt_f32_e64 s[6:7], s29, 0", 8), ("v_cndmask_b32_e64 v9, v1, v2, s[6:7]", 8), ("s_mov_b64 s[6:7], s[40:41]", 4),
            ("global_load_dwordx4 v[22:25], v18, s[6:7]", 8), ("s_endpgm", 4)]
    assert chk.hazard_scan("k", listing(rows)) == []
    # ... but only for the registers it defines
    rows[2] = ("s_mov_b32 s6, s40", 4)
    assert len(chk.hazard_scan("k", listing(rows))) == 1


# ---- MFMA results touched too early (asm-issued MFMAs: the compiler pads nothing) -------------------------------------------------
def test_a_valu_read_of_an_mfma_result_needs_eleven_wait_states():
    mf = ("v_mfma_f32_32x32x16_bf16 v[0:15], a[0:3], v[20:23], v[0:15]", 8)
    early = [mf, ("s_nop 7", 4), ("v_sin_f32_e32 v30, v4", 4), ("s_endpgm", 4)]                  # 8 wait states
    found = chk.mfma_result_scan("k", listing(early))
    assert len(found) == 1 and found[0][3] == 8
    ok = [mf, ("s_nop 7", 4), ("s_nop 2", 4), ("v_sin_f32_e32 v30, v4", 4), ("s_endpgm", 4)]     # 11
    assert chk.mfma_result_scan("k", listing(ok)) == []
    # accumulating into the same registers is what MFMAs do back to back; reading the result as an A / B operand is not
    chain = [mf, mf, ("v_mfma_f32_32x32x16_bf16 v[40:55], v[0:3], v[20:23], v[40:55]", 8), ("s_endpgm", 4)]
    found = chk.mfma_result_scan("k", listing(chain))
    assert len(found) == 1 and "v[40:55], v[0:3]" in found[0][1]
    # an overwrite (bias tile read from LDS into the accumulator) counts as a touch too
    over = [mf, ("ds_read_b128 v[0:3], v60", 8), ("s_endpgm", 4)]
    assert len(chk.mfma_result_scan("k", listing(over))) == 1
