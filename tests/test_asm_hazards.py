"""Static hazard check of the library's gfx950 code (tools/asm_hazard_check.py): no vector register that an in-flight
global / scratch / buffer load will write is read before an `s_waitcnt vmcnt` covers that load.  hipcc guarantees that for
its own loads; for the loads the kernels issue from inline asm (rk_dma.hpp fin_load, rk3d_slab.hip load_f1 / load_x2 / x4)
it holds only if the compiler neither copies nor spills the destination between the asm load and the asm wait -- which is
what broke every fused d(shift) in round 6 and what made the (since deleted) ring-of-3 slab backward spill stale x values.
Runs on the build box: no GPU needed, ~35 s."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_hazard_check as hz  # noqa: E402


def _insts(text):
    out = []
    for addr, line in enumerate(text.strip().splitlines()):
        mn, _, ops = line.strip().partition(" ")
        out.append((4 * addr, mn, ops.strip()))
    return out


def test_the_checker_sees_the_round6_defect_and_accepts_the_fix():
    broken = _insts("""
        global_load_dwordx4 v[2:5], v[24:25], off sc1
        s_nop 0
        v_mov_b32_e32 v10, v2
        s_waitcnt vmcnt(0)
    """)
    found = hz.check_function("broken", broken)
    assert len(found) == 1 and found[0][1].startswith("v_mov_b32_e32 v10, v2") and found[0][4] == "v2"
    fixed = _insts("""
        global_load_dwordx4 v[2:5], v[24:25], off sc1
        global_load_dwordx4 v[6:9], v[26:27], off sc1
        s_waitcnt vmcnt(0)
        v_mov_b32_e32 v10, v2
        v_mov_b32_e32 v11, v6
    """)
    assert hz.check_function("fixed", fixed) == []


def test_counted_waits_and_block_boundaries():
    # in-order model: vmcnt(1) covers everything but the newest operation -- DMA and stores count, have no destination
    ok = _insts("""
        global_load_dwordx2 v[2:3], v0, s[18:19] nt
        global_load_lds_dwordx4 v8, s[4:5] nt
        s_waitcnt vmcnt(1)
        v_add_f32_e32 v4, v2, v3
    """)
    assert hz.check_function("ok", ok) == []
    short = _insts("""
        global_load_dwordx2 v[2:3], v0, s[18:19] nt
        global_load_lds_dwordx4 v8, s[4:5] nt
        global_store_dword v[10:11], v12, off
        s_waitcnt vmcnt(3)
        v_add_f32_e32 v4, v2, v3
    """)
    assert len(hz.check_function("short", short)) == 1          # three operations issued, three allowed out: the load may still be one
    assert hz.check_function("exact", [i if i[1] != "s_waitcnt" else (i[0], i[1], "vmcnt(2)") for i in short]) == []
    spill = _insts("""
        global_load_dwordx2 v[2:3], v0, s[18:19] nt
        scratch_store_dwordx2 off, v[2:3], off offset:16
    """)
    assert len(hz.check_function("spill", spill)) == 1          # the deleted slab instantiation's defect
    # a masked second write of the same register is compiler output, not a hazard; a branch ends the replay
    masked = _insts("""
        global_load_dwordx4 v[8:11], v[50:51], off
        global_load_dword v8, v[50:51], off
        s_cbranch_execz 3
        v_mov_b32_e32 v1, v8
    """)
    assert hz.check_function("masked", masked) == []


def test_librubiks_hip_has_no_load_use_hazard():
    lib = os.path.join(ROOT, "rubiksnet_amd", "csrc", "librubiks_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(hz.LLVM, "llvm-objdump")):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    report, nfun, nload = hz.check_library(lib)
    assert nfun > 500 and nload > 10000, (nfun, nload)           # every code object of the library was really walked
    assert report == [], "\n".join("%s @%x: %s (%s of %x: %s)" % (r[0][:90], r[1], r[2], r[5], r[3], r[4]) for r in report[:10])
