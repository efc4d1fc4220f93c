#!/usr/bin/env python3
"""Build check for the inline-asm gathers of the pipelined march kernels (csrc/brats_device.h: async_load_vec4).

The compiler does not know those instructions are loads, so nothing but the kernels' own structure keeps a gather's
destination registers untouched between the gather and the s_waitcnt that retires it.  This script disassembles the built
library and, for every non-skipping brats_march_pipe_kernel, walks the code in layout order with a FIFO of outstanding vector
loads (retired in order by every `s_waitcnt vmcnt(N)`): any instruction that reads OR writes a register of an outstanding
load — a use scheduled above the wait, a copy, a spill — is reported and the exit status is 1.  (Layout order stands in for
control flow: the loop bodies of these kernels are straight-line between their waits, and a violation on any path shows up
as a violation in the listing.)

    python3 tools/check_async_loads.py [path/to/libmrirt.so]
"""
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
LOAD = re.compile(r"^\s*(global_load_dword(?:x[234])?)\s+(v\[\d+:\d+\]|v\d+)\s*,")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)")
BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(-?\d+)")


def parse(lines):
    """[(line_no, addr, size, instruction text)] of one function"""
    out = []
    for no, text in lines:
        m = ADDR.search(text)
        ins = text.split("//")[0].strip()
        if not m or not ins:
            continue
        out.append((no, int(m.group(1), 16), 4 * len(m.group(2).split()), ins))
    return out


def merge(a, b):
    """position-wise union of two pending lists (oldest first), aligned at their NEWEST ends"""
    n = max(len(a), len(b))
    pa = (frozenset(),) * (n - len(a)) + tuple(a)
    pb = (frozenset(),) * (n - len(b)) + tuple(b)
    return tuple(x | y for x, y in zip(pa, pb))


def check(name, lines):
    """Forward data-flow over the function's control-flow graph.  State = the gathers that may still be in flight, oldest
    first (their destination registers); `s_waitcnt vmcnt(N)` keeps the newest N; at a join the states are united position
    by position from the newest end.  Any other instruction touching a register of the state is a violation."""
    ins = parse(lines)
    if not ins:
        return []
    index = {addr: i for i, (_, addr, _, _) in enumerate(ins)}
    leaders = {0}
    succ_of = {}
    for i, (_, addr, size, text) in enumerate(ins):
        m = BRANCH.match(text)
        if m:
            tgt = index.get(addr + size + 4 * int(m.group(2)))
            nxt = i + 1 if i + 1 < len(ins) else None
            succ_of[i] = ([tgt] if tgt is not None else []) + ([nxt] if m.group(1) != "s_branch" and nxt is not None else [])
            if tgt is not None:
                leaders.add(tgt)
            if nxt is not None:
                leaders.add(nxt)
        elif text.startswith("s_endpgm"):
            succ_of[i] = []
            if i + 1 < len(ins):
                leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        blocks.append((st, en))
        block_of[st] = bi
    succs = []
    for st, en in blocks:
        last = en - 1
        if last in succ_of:
            succs.append([block_of[t] for t in succ_of[last] if t in block_of])
        else:
            succs.append([block_of[en]] if en in block_of else [])

    def transfer(bi, state, report):
        st, en = blocks[bi]
        pend = list(state)
        for i in range(st, en):
            no, _, _, text = ins[i]
            w = WAIT.match(text)
            if w:
                m = re.search(r"vmcnt\((\d+)\)", w.group(1))
                if m:
                    pend = pend[len(pend) - int(m.group(1)):] if int(m.group(1)) < len(pend) else pend
                    if int(m.group(1)) == 0:
                        pend = []
                continue
            l = LOAD.match(text)
            ops = text.split(None, 1)[1] if " " in text else ""
            touched = regs(ops)
            if l:
                dest = regs(l.group(2))
                touched = regs(ops.split(",", 1)[1]) if "," in ops else set()       # the address operand; the destination is checked below
                touched |= dest
            flying = set().union(*pend) if pend else set()
            if report is not None and touched & flying:
                report.append((no, text))
            if l:
                pend.append(frozenset(regs(l.group(2))))
                pend = pend[-64:]
        return tuple(pend)

    instate = {0: ()}
    work = [0]
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        bi = work.pop()
        out = transfer(bi, instate[bi], None)
        for sb in succs[bi]:
            new = out if sb not in instate else merge(instate[sb], out)
            if sb not in instate or new != instate[sb]:
                instate[sb] = new
                work.append(sb)
    bad = []
    for bi in sorted(instate):
        transfer(bi, instate[bi], bad)
    return bad


# ---- the gfx9 hazard "a VALU instruction writes an SGPR, a vector-memory instruction reads it fewer than 5 wait states later" ----
# The hardware does not interlock it; the compiler pads loads IT issues with s_nop, but not inline asm (LLVM's hazard recogniser
# does not look inside asm statements).  The asm gathers therefore copy their base pointer with an s_mov_b64 inside the
# statement (csrc/brats_device.h); this scan proves, on every kernel of the built library, that no vector-memory
# instruction is left reading a VALU-written SGPR too early — e.g. a spilled base pointer restored with v_readlane_b32 in
# front of a gather, which is what faulted once in round 3 and once in round 4.
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)\w+\s+(.*)$")
HAZARD_WAIT_STATES = 5


def sregs(tok):
    out = set()
    for m in SREG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    if "vcc" in tok:
        out.update((106, 107))
    return out


def valu_sgpr_writes(text):
    """SGPRs a VALU instruction writes: the destination of v_readlane / v_readfirstlane / VOP3 compares, the carry-out of
    v_div_scale / v_add_co & co, VCC for the e32 compares."""
    mn = text.split()[0]
    if not mn.startswith("v_"):
        return set()
    ops = text.split(None, 1)[1].split(",") if " " in text else []
    w = sregs(ops[0]) if ops else set()
    if mn.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
        w |= sregs(ops[1])
    if mn.startswith("v_cmp") and ops and not re.match(r"^\s*(s\[|s\d|vcc)", ops[0]):
        w = {106, 107}
    return w


SALU_NO_DST = ("s_cmp", "s_bitcmp", "s_setprio", "s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "s_sleep",
               "s_sendmsg", "s_store", "s_dcache", "s_icache", "s_setreg", "s_sethalt", "s_trap", "s_inst_prefetch", "s_clause",
               "s_setpc", "s_swappc", "s_rfe", "s_ttrace", "s_incperflevel", "s_decperflevel", "s_set_gpr_idx", "s_atc_probe",
               "s_scratch_store", "s_buffer_store")


def scalar_sgpr_writes(text):
    """SGPRs a SCALAR instruction (SALU / SMEM load) defines: its first operand.  A vector-memory read of an SGPR a scalar
    instruction wrote last has no hazard, whatever a VALU instruction wrote into it earlier."""
    mn = text.split()[0]
    if not mn.startswith("s_") or mn.startswith(SALU_NO_DST) or " " not in text:
        return set()
    return sregs(text.split(None, 1)[1].split(",")[0])


def hazard_scan(name, lines):
    """[(line_no, vmem instruction, writer instruction, wait states)] for every vector-memory instruction that reads an SGPR a
    VALU instruction wrote fewer than HAZARD_WAIT_STATES wait states earlier, on ANY path through the function."""
    ins = parse(lines)
    if not ins:
        return []
    index = {addr: i for i, (_, addr, _, _) in enumerate(ins)}
    preds = {i: [] for i in range(len(ins))}
    for i, (_, addr, size, text) in enumerate(ins):
        m = BRANCH.match(text)
        falls = not (text.startswith("s_endpgm") or (m and m.group(1) == "s_branch"))
        if falls and i + 1 < len(ins):
            preds[i + 1].append(i)
        if m:
            off = int(m.group(2))
            off = off - 65536 if off >= 32768 else off
            tgt = index.get(addr + size + 4 * off)
            if tgt is not None:
                preds[tgt].append(i)
    found = []
    for i, (no, _, _, text) in enumerate(ins):
        m = VMEM.match(text)
        if not m:
            continue
        need = sregs(m.group(2))
        if not need:
            continue
        # walk backwards over every path until HAZARD_WAIT_STATES wait states have been seen; registers a scalar instruction
        # defined on the way are no longer looked for on that path
        stack, seen, hit = [(p, 0, frozenset(need)) for p in preds[i]], set(), None
        while stack and hit is None:
            j, ws, want = stack.pop()
            if (j, ws, want) in seen or ws >= HAZARD_WAIT_STATES or not want:
                continue
            seen.add((j, ws, want))
            t = ins[j][3]
            if valu_sgpr_writes(t) & want:
                hit = (no, text, t, ws)
                break
            want = want - scalar_sgpr_writes(t)
            step = int(t.split()[1]) + 1 if t.startswith("s_nop") else 1       # (s_waitcnt counts as one issue slot, like any instruction)
            for p in preds[j]:
                stack.append((p, ws + step, want))
        if hit:
            found.append(hit)
    return found


# ---- matrix-core results: an MFMA's destination may not be touched by anything but an accumulating MFMA too early -----------------
# The weight-stationary and refinement INR kernels issue their MFMAs as inline asm (AGPR-pinned operands), so the compiler's own
# padding between an MFMA and the first VALU / LDS / vector-memory instruction that reads or overwrites its result (gfx940 family:
# passes + 3 wait states — 11 for the 8-pass 32x32x16 bf16 product; hipcc keeps 12 in the kernels it schedules itself) is the
# kernel author's job there.  Same lesson as the SGPR hazard above: what the compiler cannot see, the build checks.  Counted
# as LLVM counts them (one per instruction, N + 1 per s_nop N), in layout order inside a function.
VREG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
MFMA_RESULT_WAIT_STATES = 11


def vregs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def mfma_result_scan(name, lines):
    """[(line_no, instruction, mfma, wait states)]: instructions other than an MFMA that names the same registers as its C/D
    accumulator which read or write an MFMA's destination fewer than MFMA_RESULT_WAIT_STATES wait states after it."""
    ins = parse(lines)
    last, found, states = {}, [], 0
    at = []
    for _, _, _, text in ins:
        at.append(states)
        states += int(text.split()[1]) + 1 if text.startswith("s_nop") else 1
    for i, (no, _, _, text) in enumerate(ins):
        mn = text.split()[0]
        ops = text.split(None, 1)[1] if " " in text else ""
        parts = [p.strip() for p in ops.split(",")]
        if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
            dst = vregs(parts[0])
            for p in parts[1:3]:                                   # A / B operands read another MFMA's result
                for r in vregs(p):
                    if r in last and at[i] - at[last[r]] - 1 < MFMA_RESULT_WAIT_STATES:
                        found.append((no, text, ins[last[r]][3], at[i] - at[last[r]] - 1))
                        break
            for r in dst:
                last[r] = i
            continue
        if mn.startswith("s_") or not ops:
            continue
        touched = [r for r in vregs(ops) if r in last]
        if touched:
            gap = min(at[i] - at[last[r]] - 1 for r in touched)
            if gap < MFMA_RESULT_WAIT_STATES:
                found.append((no, text, ins[last[touched[0]]][3], gap))
        for r in vregs(parts[0]):                                  # a register redefined by something else is no MFMA result any more
            last.pop(r, None)
    return found


TOOLS = ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


class ToolsMissing(RuntimeError):
    pass


def llvm_bin() -> pathlib.Path:
    """The ROCm LLVM tool directory of the compiler that built the library: $MRIRT_LLVM_BIN, else next to $HIPCC
    (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), else $ROCM_PATH, else /opt/rocm."""
    import os
    import shutil
    cands = []
    if os.environ.get("MRIRT_LLVM_BIN"):
        cands.append(pathlib.Path(os.environ["MRIRT_LLVM_BIN"]))
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        cands.append(pathlib.Path(hipcc).resolve().parent.parent / "lib" / "llvm" / "bin")
    if os.environ.get("ROCM_PATH"):
        cands.append(pathlib.Path(os.environ["ROCM_PATH"]) / "lib" / "llvm" / "bin")
    cands.append(pathlib.Path("/opt/rocm/lib/llvm/bin"))
    for c in cands:
        if all((c / t).exists() for t in TOOLS):
            return c
    raise ToolsMissing("check_async_loads: need " + ", ".join(TOOLS) + " of the ROCm LLVM that built the library; looked in "
                       + ", ".join(str(c) for c in cands) + " — set MRIRT_LLVM_BIN to the directory that holds them")


def device_disassembly(so: pathlib.Path) -> str:
    """gfx950 disassembly of every code object embedded in the library: the .hip_fatbin section holds one offload bundle per
    HIP source; each is unbundled and disassembled."""
    import tempfile
    LLVM = llvm_bin()
    out = []
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        fat = td / "fatbin.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), str(fat)], check=True)
        blob = fat.read_bytes()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, st in enumerate(starts):
            chunk = td / f"bundle{i}.bin"
            chunk.write_bytes(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = td / f"dev{i}.co"
            r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={chunk}", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not co.exists() or co.stat().st_size == 0:
                continue
            out.append(subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], capture_output=True, text=True).stdout)
    return "\n".join(out)


def main():
    so = pathlib.Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "mri-raytracer_amd" / "libmrirt.so"
    try:
        text = device_disassembly(so)
    except ToolsMissing as e:
        print(e)
        return 3
    if "s_endpgm" not in text or "brats_march_pipe_kernel" not in text:
        print(f"{so}: no device disassembly of brats_march_pipe_kernel found")
        return 2
    funcs, cur = {}, None
    for no, line in enumerate(text.splitlines()):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append((no, line))                 # "\t<insn> operands   // addr: encoding"
    checked = failures = loads = 0
    for name, lines in funcs.items():
        if "brats_march_pipe_kernel" not in name or name.startswith("__"):
            continue
        # template arguments ...ELb<SKIP>EE: the skipping kernels use compiler-visible loads
        m = re.search(r"pipe_kernelILb[01]ELi\d+ELb[01]ELi\d+ELb[01]ELb[01]ELb([01])E", name)
        if not m or m.group(1) == "1":
            continue
        checked += 1
        loads += sum(1 for _, t in lines if LOAD.match(t.split("//")[0].strip()))
        bad = check(name, lines)
        if bad:
            failures += 1
            print(f"FAIL {name}: {len(bad)} access(es) to an in-flight gather destination")
            for no, ins in bad[:6]:
                print(f"   line {no}: {ins}")
    print(f"check_async_loads: {checked} kernels, {loads} gathers checked, {failures} kernels failing")
    hz_kernels = hz = vm = 0
    for name, lines in funcs.items():
        if name.startswith("__"):
            continue
        vm += sum(1 for _, t in lines if VMEM.match(t.split("//")[0].strip()))
        bad = hazard_scan(name, lines)
        if bad:
            hz_kernels += 1
            hz += len(bad)
            print(f"HAZARD {name}: {len(bad)} vector-memory instruction(s) read an SGPR a VALU instruction wrote < {HAZARD_WAIT_STATES} wait states earlier")
            for no, ins, writer, ws in bad[:6]:
                print(f"   line {no}: {writer}   ->   {ins}   ({ws} wait state(s) between)")
    print(f"check_async_loads: VALU-writes-SGPR -> VMEM hazard: {len(funcs)} kernels, {vm} vector-memory instructions, {hz} violation(s) in {hz_kernels} kernel(s)")
    mf_kernels = mf = mfmas = 0
    for name, lines in funcs.items():
        if name.startswith("__"):
            continue
        n = sum(1 for _, t in lines if t.split("//")[0].strip().startswith("v_mfma"))
        if not n:
            continue
        mfmas += n
        bad = mfma_result_scan(name, lines)
        if bad:
            mf_kernels += 1
            mf += len(bad)
            print(f"MFMA {name}: {len(bad)} instruction(s) touch an MFMA result < {MFMA_RESULT_WAIT_STATES} wait states after it")
            for no, ins, producer, ws in bad[:6]:
                print(f"   line {no}: {producer[:60]}   ->   {ins}   ({ws} wait state(s) between)")
    print(f"check_async_loads: MFMA result hazard: {mfmas} MFMAs, {mf} violation(s) in {mf_kernels} kernel(s)")
    return 1 if failures or hz or mf or not checked or not loads else 0


if __name__ == "__main__":
    sys.exit(main())
