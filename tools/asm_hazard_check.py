#!/usr/bin/env python
"""Static check of the gfx950 code in librubiks_hip.so for the hazard that produced the round-6 finalizer defect:
a vector register that is the destination of a global / scratch / buffer LOAD still in flight is read
before an `s_waitcnt vmcnt(N)` has covered that load.

hipcc inserts such waits for the loads it emits itself.  For loads issued from inline asm it cannot: to the compiler an asm
statement is an ordinary instruction whose outputs are ready when the statement ends, so it may copy or reuse the destination
registers right behind it.  The library issues loads from asm in a few places on purpose (LDS-DMA, and register loads that must
not drain the DMA queue: rk_dma.hpp `fin_load`, rk3d_slab.hip `load_f1` / `load_x2`), with hand-counted waits.  This script
disassembles every code object in the library and replays each BASIC BLOCK with the in-order model the kernels rely on
(VMEM operations retire in issue order; `vmcnt(N)` = at most N still outstanding): a use of a pending destination inside the
block that issued the load is a defect, whatever the register allocator did.  Loads still pending at the end of a block are
not followed across the branch (no false positives; the blocks of the hand-pipelined walks are long straight-line code).

    python tools/asm_hazard_check.py [path/to/librubiks_hip.so]      # exit status 1 and a listing when something is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")
_STORE_LIKE = ("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write", "ds_store", "global_atomic",
               "buffer_atomic", "flat_atomic", "ds_add", "ds_max", "ds_min", "ds_swizzle")
_VMEM = ("global_load", "global_store", "global_atomic", "scratch_load", "scratch_store", "buffer_load", "buffer_store",
         "buffer_atomic", "flat_load", "flat_store", "flat_atomic")
_ACCUMULATING = ("fmac", "_mac_", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c")


def code_objects(lib):
    """Every gfx950 code object bundled in the shared library, as temporary files."""
    data = open(lib, "rb").read()
    starts = [m.start() for m in re.finditer(MAGIC, data)]
    out = []
    tmp = tempfile.mkdtemp(prefix="rk_hazard_")
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(data)
        blob = os.path.join(tmp, "bundle%d" % i)
        open(blob, "wb").write(data[s:e])
        co = os.path.join(tmp, "co%d" % i)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + blob,
                            "--targets=" + TARGET, "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def regs(text):
    s = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def split_operands(ops):
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def check_function(name, insts):
    """insts: list of (addr, mnemonic, operand string).  Returns hazards as (addr, text, load addr, load text)."""
    targets = set()
    for addr, mn, ops in insts:
        if mn.startswith("s_cbranch") or mn == "s_branch":
            try:
                imm = int(ops.split()[0])
            except (ValueError, IndexError):
                continue
            if imm >= 32768:
                imm -= 65536
            targets.add(addr + 4 + 4 * imm)
    hazards, pending = [], []                      # pending: (dest regs, addr, text) of VMEM ops in issue order
    for addr, mn, ops in insts:
        if addr in targets:
            pending = []
        operands = split_operands(ops)
        is_vmem = mn.startswith(_VMEM)
        is_dma = "_lds_" in mn or mn.endswith("_lds") or " lds" in (" " + ops)
        writes_first = bool(operands) and _VREG.fullmatch(operands[0]) is not None and not mn.startswith(_STORE_LIKE) \
            and not (is_vmem and is_dma)
        dest = regs(operands[0]) if writes_first else set()
        srcs = set()
        for k, op in enumerate(operands):
            if k == 0 and writes_first and not any(t in mn for t in _ACCUMULATING):
                continue
            srcs |= regs(op)
        if pending:
            busy = {}
            for d, a, t in pending:
                for r in d:
                    busy[r] = (a, t)
            # READS only: a write to a pending destination is legitimate compiler output under disjoint EXEC masks (`v = cond ?
            # load : 0` becomes a masked load and a masked v_mov / second load of the same register; returns are in order)
            for r in sorted(srcs & set(busy)):
                hazards.append((addr, "%s %s" % (mn, ops), busy[r][0], busy[r][1], "v%d" % r, "read"))
                break
        if mn == "s_waitcnt":
            m = _VMCNT.search(ops)
            if m:
                n = int(m.group(1))
                pending = pending[-n:] if n > 0 else []
        elif is_vmem:
            is_load = "_load" in mn and not is_dma
            pending.append((dest if is_load else set(), addr, "%s %s" % (mn, ops)))
        if mn.startswith("s_cbranch") or mn in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            pending = []
    return hazards


def check_library(lib):
    report, nfun, nload = [], 0, 0
    for co in code_objects(lib):
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True,
                             text=True, check=True).stdout
        name, insts = None, []

        def flush():
            nonlocal nfun, nload
            if name and insts:
                nfun += 1
                nload += sum(1 for _, mn, ops in insts if mn.startswith(("global_load", "scratch_load", "buffer_load"))
                             and "_lds_" not in mn)
                for h in check_function(name, insts):
                    report.append((name,) + h)

        for line in asm.splitlines():
            f = _FUNC.match(line)
            if f:
                flush()
                name, insts = f.group(1), []
                continue
            m = _LINE.match(line)
            if m and name:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
        flush()
    return report, nfun, nload


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rubiksnet_amd", "csrc", "librubiks_hip.so")
    report, nfun, nload = check_library(lib)
    print("%s: %d kernels / device functions, %d register loads replayed, %d hazard(s)" % (lib, nfun, nload, len(report)))
    for name, addr, text, laddr, ltext, reg, how in report[:40]:
        print("  %s\n    %06x: %s   -- %s %s while its load is in flight:\n    %06x: %s" % (name[:110], addr, text, reg, how, laddr, ltext))
    return 1 if report else 0


if __name__ == "__main__":
    sys.exit(main())
