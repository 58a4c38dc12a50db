"""gfx950 hazard lint for hand-written and generated machine code (CPU only, no GPU needed).

The hardware does not interlock a handful of dependencies: the assembler does not insert the wait states either, and a
compiler would (LLVM's GCNHazardRecognizer).  This module walks a disassembly (llvm-objdump -d) along its control flow and
checks the gfx90a / gfx940-family table for the instruction classes this repository emits:

    R1  VALU writes an SGPR / VCC          -> VALU reads it (operand, carry-in, v_cndmask's VCC)      2 wait states
    R2  VALU writes an SGPR / VCC          -> vector memory instruction reads it (saddr, soffset)      5
    R3  VALU writes an SGPR / VCC          -> v_readlane / v_writelane lane select                     4
    R4  VALU writes VCC                    -> v_div_fmas                                               4
    R5  SALU writes M0 (s_mov m0, s_set_gpr_idx_on / _idx / _mode) -> s_movrel*, ds_*_addtid, GDS, s_sendmsg, `lds` loads   1
    R6  VALU writes EXEC (v_cmpx)          -> v_readlane / v_readfirstlane / v_writelane               4
    R7  VALU writes a VGPR                 -> v_readlane / v_readfirstlane reads it                    1
    R8  VALU writes a VGPR                 -> vector store of more than 64 bits reads it as data       2
    R9  VALU writes EXEC                   -> DPP instruction                                          5   (none emitted; flagged)
    R10 transcendental VALU result         -> next VALU reads it                                       1

plus the calling convention of the interpreter's VGPR index mode (fx_interp_handlers.inc): while s_set_gpr_idx_on is in
force, the operand position it makes M0-relative may hold only the register-file base (v32), an SGPR or a constant, the
other positions only plain registers below v32 - and a handler must set or clear the mode before its first VALU instruction,
because it inherits whatever the previous handler left (DST mode after a saturating store).  A plain VGPR in a relative
position silently addresses v(n + M0): a wrong row, or v27 - the byte offset of every memory access of the lane.

A wait state = one instruction issued in between (s_nop N counts N + 1).  Paths are followed backwards through fall-through
and branch edges; at an indirect entry (a label reached by s_setpc_b64, the instruction after a call) the unknown predecessor
is assumed to have written every SGPR that any VALU instruction of the image writes, one wait state (the s_setpc) ago.
"""
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
RF_BASE = 32

_SREG = re.compile(r"^(?:s(\d+)|s\[(\d+):(\d+)\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|flat_scratch|ttmp\d+|ttmp\[\d+:\d+\])$")
_VREG = re.compile(r"^(?:v(\d+)|v\[(\d+):(\d+)\])$")
_TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
_VMEM = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_", "image_")


class Ins:
    __slots__ = ("addr", "size", "mnem", "ops", "label", "text", "index", "mods", "word")

    def __repr__(self):
        return "%06x %s" % (self.addr, self.text)


def _strip(op):
    """operand without source modifiers: -v3, |v3|, -|v3|, neg(v3), abs(v3), sext(v3)"""
    op = op.strip()
    m = re.match(r"^-?\|(.+)\|$", op)
    if m:
        return m.group(1)
    m = re.match(r"^(?:neg|abs|sext)\((.+)\)$", op)
    if m:
        return _strip(m.group(1))
    if op.startswith("-") and _VREG.match(op[1:]) or op.startswith("-") and _SREG.match(op[1:]):
        return op[1:]
    return op


def sregs_of(op):
    """set of scalar register names an operand names: s<k>, 'vcc', 'exec', 'm0'"""
    op = _strip(op)
    m = _SREG.match(op)
    if not m:
        return set()
    if m.group(1) is not None:
        return {"s%d" % int(m.group(1))}
    if m.group(2) is not None:
        return {"s%d" % k for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    if op.startswith("vcc"):
        return {"vcc"}
    if op.startswith("exec"):
        return {"exec"}
    if op == "m0":
        return {"m0"}
    return {op}


def vregs_of(op):
    op = _strip(op)
    m = _VREG.match(op)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def parse_objdump(text):
    out, label = [], None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            label = m.group(1)
            continue
        m = re.match(r"^\t(\S+)(?:\s+(.*?))?\s*// ([0-9A-F]+): ((?:[0-9A-F]{8} ?)+)", line)
        if not m:
            continue
        i = Ins()
        i.mnem = m.group(1)
        body = (m.group(2) or "").strip()
        i.text = (i.mnem + " " + body).strip()
        # operands are separated by commas outside brackets / parentheses; trailing modifiers ("off", "offset:8", "nt") are separated by spaces
        ops, depth, cur = [], 0, ""
        for ch in body:
            if ch in "[(":
                depth += 1
            elif ch in "])":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
        i.mods = []
        if ops:
            parts = ops[-1].split()
            # "v7, s[36:37] offset:16 nt" -> last operand "s[36:37]", modifiers after it; a lone modifier operand ("off") stays an operand
            if len(parts) > 1 and not parts[0].startswith(("gpr_idx", "vmcnt", "lgkmcnt", "expcnt", "hwreg", "sendmsg")):
                ops[-1] = parts[0]
                i.mods = parts[1:]
        i.ops = ops
        i.addr = int(m.group(3), 16)
        i.size = 4 * len(m.group(4).split())
        i.word = int(m.group(4).split()[0], 16)
        i.label = label
        label = None
        i.index = len(out)
        out.append(i)
    return out


def drop_data(ins, data_suffixes=("_table", "_hole")):
    """removes what objdump decoded from data symbols (the handler offset table, the s_nop filler of the code hole)"""
    out, skipping = [], False
    for i in ins:
        if i.label is not None:
            skipping = i.label.endswith(tuple(data_suffixes))
        if not skipping:
            i.index = len(out)
            out.append(i)
    return out


def disassemble_object(path):
    return drop_data(parse_objdump(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", path], check=True, capture_output=True, text=True).stdout))


def disassemble_listing(listing):
    """llvm-mc syntax listing -> instruction list (assembled for gfx950 and disassembled again: canonical text, addresses)"""
    with tempfile.TemporaryDirectory() as d:
        src, obj = os.path.join(d, "x.s"), os.path.join(d, "x.o")
        with open(src, "w") as fh:
            fh.write(".text\n" + listing + "\n")
        subprocess.run([os.path.join(LLVM, "llvm-mc"), "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", obj, src], check=True, capture_output=True)
        return disassemble_object(obj)


# ---------------------------------------------------------------------------------------------------------------- classes
def is_valu(i):
    return i.mnem.startswith("v_")


def is_salu(i):
    return i.mnem.startswith("s_") and not is_smem(i)


def is_smem(i):
    return i.mnem.startswith(("s_load_", "s_store_", "s_buffer_", "s_scratch_", "s_dcache", "s_atc_", "s_memtime", "s_memrealtime", "s_atomic_"))


def is_vmem(i):
    return i.mnem.startswith(_VMEM)


def is_ds(i):
    return i.mnem.startswith("ds_")


def wait_states(i):
    if i.mnem == "s_nop":
        return int(i.ops[0], 0) + 1
    return 1


def is_branch(i):
    return i.mnem == "s_branch" or i.mnem.startswith("s_cbranch_")


def ends_flow(i):
    return i.mnem in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_rfe_b64")


def branch_target(i):
    simm = i.word & 0xFFFF   # (objdump prints a symbol instead of the number where a relocation applies)
    if simm >= 32768:
        simm -= 65536
    return i.addr + 4 + 4 * simm


_CARRY_OUT = re.compile(r"^v_(?:add|sub|subrev|addc|subb|subbrev)_co_(?:u32|ci_u32)|^v_div_scale_|^v_mad_[ui]64_[ui]32")


def valu_sgpr_defs(i):
    """scalar registers a VALU instruction writes"""
    m = i.mnem
    if m.startswith("v_cmpx"):
        d = {"exec"}
        if i.ops and sregs_of(i.ops[0]):  # e64 form with an explicit sdst on some targets
            d |= sregs_of(i.ops[0])
        return d
    if m.startswith("v_cmp"):
        return sregs_of(i.ops[0])
    if m.startswith(("v_readfirstlane", "v_readlane")):
        return sregs_of(i.ops[0])
    if _CARRY_OUT.match(m):
        return sregs_of(i.ops[1])
    return set()


def valu_src_ops(i):
    """source operands of a VALU instruction (everything but the destinations)"""
    m = i.mnem
    if _CARRY_OUT.match(m):
        return i.ops[2:]
    return i.ops[1:]


def valu_sgpr_uses(i):
    u = set()
    for op in valu_src_ops(i):
        u |= sregs_of(op)
    u.discard("exec")  # every VALU instruction is masked by EXEC: interlocked
    return u


def valu_vgpr_defs(i):
    m = i.mnem
    if m.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
        return set()
    return vregs_of(i.ops[0]) if i.ops else set()


_SALU_NO_DST = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_setpc", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_set_gpr_idx",
                "s_setprio", "s_sendmsg", "s_setreg", "s_icache", "s_dcache", "s_setvskip", "s_sethalt", "s_trap", "s_incperflevel", "s_decperflevel", "s_ttrace")


def salu_sgpr_defs(i):
    """scalar registers a SALU / SMEM instruction overwrites (it reads its sources interlocked, so a later VALU read sees ITS value)"""
    if is_smem(i):
        return sregs_of(i.ops[0]) if i.mnem.startswith(("s_load_", "s_buffer_load", "s_memtime", "s_memrealtime")) and i.ops else set()
    if not is_salu(i) or i.mnem.startswith(_SALU_NO_DST) or not i.ops:
        return set()
    d = sregs_of(i.ops[0])
    if "saveexec" in i.mnem:
        d |= {"exec"}
    return d


def m0_writers(i):
    if i.mnem in ("s_set_gpr_idx_on", "s_set_gpr_idx_idx", "s_set_gpr_idx_mode"):
        return True
    return is_salu(i) and bool(i.ops) and "m0" in sregs_of(i.ops[0]) and not i.mnem.startswith(("s_cmp", "s_bitcmp", "s_cbranch", "s_setpc", "s_waitcnt"))


def m0_readers(i):
    if i.mnem.startswith(("s_movrel", "s_sendmsg", "v_interp", "ds_gws", "v_movrel")):
        return True
    if is_ds(i) and ("addtid" in i.mnem or "gds" in i.mods):
        return True
    if is_vmem(i) and "lds" in i.mods:
        return True
    return False


def store_data_regs(i):
    """VGPRs a vector store of more than 64 bits reads as data"""
    if not is_vmem(i) or "store" not in i.mnem:
        return set()
    if not re.search(r"(dwordx3|dwordx4|_xyz|_xyzw|b96|b128)$", i.mnem):
        return set()
    # global_store_dwordx4 vaddr, vdata, saddr   /   buffer_store_dwordx4 vdata, vaddr, srsrc, soffset
    data = i.ops[1] if i.mnem.startswith(("global_", "flat_", "scratch_")) else i.ops[0]
    return vregs_of(data)


class Finding:
    def __init__(self, rule, use, dfn, have, need):
        self.rule, self.use, self.dfn, self.have, self.need = rule, use, dfn, have, need

    def __repr__(self):
        return "%s: `%s` @%06x needs %d wait state(s) after `%s` @%s, has %d" % (
            self.rule, self.use.text, self.use.addr, self.need, self.dfn.text if self.dfn else "<unknown predecessor>",
            ("%06x" % self.dfn.addr) if self.dfn else "?", self.have)


def lint_hazards(ins, entries=(), assume_entry_defs=None):
    """ins: parse_objdump() list.  entries: addresses reached by indirect jumps (besides labels and instructions after an s_setpc_b64).
    Returns a list of Finding."""
    by_addr = {i.addr: i for i in ins}
    preds = {i.index: [] for i in ins}
    indirect = set(entries)
    for k, i in enumerate(ins):
        if is_branch(i):
            t = branch_target(i)
            if t in by_addr:
                preds[by_addr[t].index].append(k)
        if not ends_flow(i) and k + 1 < len(ins):
            preds[k + 1].append(k)
        if i.mnem in ("s_setpc_b64", "s_swappc_b64") and k + 1 < len(ins):
            indirect.add(ins[k + 1].addr)   # a call's return point (or dead code: harmless)
        if i.label is not None:
            indirect.add(i.addr)            # a symbol: may be entered through a pointer
    all_valu_sdefs = set()
    for i in ins:
        if is_valu(i):
            all_valu_sdefs |= valu_sgpr_defs(i)
    if assume_entry_defs is not None:
        all_valu_sdefs = set(assume_entry_defs)
    findings = []

    def search(use, need, regs, defs_of, rule, entry_hazard, extra=None):
        """walk back from `use` over at most `need` wait states looking for an instruction with defs_of(d) & regs (and extra(d), if
        given); a SALU / SMEM instruction that overwrites a register takes it off the list (regs = None: no register involved)"""
        seen = set()
        stack = [(use.index, 0, frozenset(regs) if regs is not None else None)]
        while stack:
            k, have, live = stack.pop()
            if ins[k].addr in indirect and entry_hazard and have + 1 < need and (live is None or live & all_valu_sdefs):
                findings.append(Finding(rule, use, None, have + 1, need))   # arrived through an indirect jump: the jump was one wait state
            for p in preds[k]:
                d = ins[p]
                got = defs_of(d)
                if (got if live is None else (got & live)) and (extra is None or extra(d)):
                    if have < need:
                        findings.append(Finding(rule, use, d, have, need))
                    continue
                l2 = live
                if live is not None:
                    l2 = live - salu_sgpr_defs(d)
                    if not l2:
                        continue
                h2 = have + wait_states(d)
                if h2 < need and (p, h2, l2) not in seen:
                    seen.add((p, h2, l2))
                    stack.append((p, h2, l2))

    vs = lambda d: valu_sgpr_defs(d) if is_valu(d) else set()
    vv = lambda d: valu_vgpr_defs(d) if is_valu(d) else set()
    for i in ins:
        if is_valu(i):
            uses = valu_sgpr_uses(i)
            if uses:
                search(i, 2, uses, vs, "R1 VALU-write-SGPR -> VALU-read", True)
            if i.mnem.startswith(("v_readlane", "v_writelane")) and len(i.ops) >= 3 and sregs_of(i.ops[2]):
                search(i, 4, sregs_of(i.ops[2]), vs, "R3 VALU-write-SGPR -> lane select", True)
            if i.mnem.startswith("v_div_fmas"):
                search(i, 4, {"vcc"}, vs, "R4 VALU-write-VCC -> v_div_fmas", True)
            if i.mnem.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                search(i, 4, {"exec"}, vs, "R6 VALU-write-EXEC -> lane access", False)
            if i.mnem.startswith(("v_readlane", "v_readfirstlane")) and vregs_of(i.ops[1]):
                search(i, 1, vregs_of(i.ops[1]), vv, "R7 VALU-write-VGPR -> readlane", False)
            if "dpp" in i.mnem or any(m.startswith(("quad_perm", "row_", "wave_")) for m in i.mods):
                search(i, 5, {"exec"}, vs, "R9 VALU-write-EXEC -> DPP", False)
            srcv = set()
            for op in valu_src_ops(i):
                srcv |= vregs_of(op)
            if srcv:
                search(i, 1, srcv, vv, "R10 trans result -> VALU", False, extra=lambda d: d.mnem.startswith(_TRANS))
        if is_vmem(i):
            uses = set()
            for op in i.ops:
                uses |= sregs_of(op)
            if uses:
                search(i, 5, uses, vs, "R2 VALU-write-SGPR -> VMEM", True)
            data = store_data_regs(i)
            if data:
                search(i, 2, data, vv, "R8 VALU-write-VGPR -> wide store data", False)
        if m0_readers(i):
            search(i, 1, None, lambda d: {"m0"} if m0_writers(d) else set(), "R5 SALU-write-M0 -> M0 user", False)
    # de-duplicate (several paths to the same pair)
    uniq, seen = [], set()
    for f in findings:
        key = (f.rule, f.use.addr, f.dfn.addr if f.dfn else -1)
        if key not in seen:
            seen.add(key)
            uniq.append(f)
    return uniq


# ------------------------------------------------------------------------------------------------- VGPR index mode
OFF, SRC0, SRC1, DST, UNKNOWN = "off", "src0", "src1", "dst", "unknown"


def _mode_of(i):
    m = re.search(r"gpr_idx\(([^)]*)\)", i.text)
    names = [t for t in (m.group(1).split(",") if m else []) if t]
    if names == ["SRC0"]:
        return SRC0
    if names == ["SRC1"]:
        return SRC1
    if names == ["DST"]:
        return DST
    return "other:" + ",".join(names)


def lint_index_mode(ins, entry_state=UNKNOWN, entries=None, kernel_labels=(), any_base=False):
    """Checks the interpreter's index-mode convention over a disassembly.  Returns a list of (Ins, message).
    Labels in kernel_labels are kernel entries (the hardware starts a wave with index mode off); every other label is a
    handler entered by s_setpc_b64 with whatever mode the previous handler left (entry_state)."""
    by_addr = {i.addr: i for i in ins}
    succ = {}
    for k, i in enumerate(ins):
        s = []
        if is_branch(i):
            t = branch_target(i)
            if t in by_addr:
                s.append(by_addr[t].index)
        if not ends_flow(i) and k + 1 < len(ins):
            s.append(k + 1)
        succ[k] = s
    state = {}
    work = []
    for i in ins:
        if isinstance(entries, dict):
            if i.addr in entries:
                state[i.index] = entries[i.addr]
                work.append(i.index)
        elif (entries is None and i.label is not None) or (entries is not None and i.addr in entries):
            state[i.index] = OFF if i.label in kernel_labels else entry_state
            work.append(i.index)
    problems = []

    def transfer(i, st):
        if i.mnem == "s_set_gpr_idx_on":
            return _mode_of(i)
        if i.mnem == "s_set_gpr_idx_off":
            return OFF
        if i.mnem in ("s_set_gpr_idx_mode", "s_set_gpr_idx_idx"):
            return UNKNOWN
        return st

    while work:
        k = work.pop()
        st = transfer(ins[k], state[k])
        for n in succ[k]:
            old = state.get(n)
            new = st if old is None or old == st else UNKNOWN
            if old != new:
                state[n] = new
                work.append(n)
    for i in ins:
        st = state.get(i.index)
        if st is None or not is_valu(i):
            continue
        if i.mnem.startswith(("v_readfirstlane", "v_readlane", "v_writelane")):
            continue  # not affected by index mode
        dst = i.ops[0] if i.ops else ""
        srcs = valu_src_ops(i)
        has_vdst = bool(vregs_of(dst)) and not i.mnem.startswith("v_cmp")
        pos = {SRC0: 0, SRC1: 1}
        if st == OFF:
            continue
        if st == UNKNOWN:
            if has_vdst or any(vregs_of(o) for o in srcs):
                problems.append((i, "VALU instruction with VGPR operands while the index mode is unknown (inherited or merged)"))
            continue
        if st == DST:
            if has_vdst and min(vregs_of(dst)) != RF_BASE:
                problems.append((i, "DST index mode: destination %s is not the register-file base v%d" % (dst, RF_BASE)))
            if not has_vdst:
                problems.append((i, "DST index mode: instruction without a VGPR destination"))
            continue
        if st in pos:
            p = pos[st]
            for q, o in enumerate(srcs):
                v = vregs_of(o)
                if not v:
                    continue
                if q == p and (min(v) < RF_BASE if any_base else min(v) != RF_BASE):   # (any_base: generated code also indexes its PCM input ring)
                    problems.append((i, "%s index mode: source %d is the plain register %s (would address v(n + M0))" % (st.upper(), q, o)))
                if q != p and min(v) >= RF_BASE and not any_base:
                    problems.append((i, "%s index mode: source %d names the register file (%s) but is not M0-relative" % (st.upper(), q, o)))
            if has_vdst and min(vregs_of(dst)) >= RF_BASE and not any_base:
                problems.append((i, "%s index mode: destination %s is inside the register file" % (st.upper(), dst)))
            continue
        problems.append((i, "index mode %s is not one the handlers use" % st))
    return problems


# ------------------------------------------------------------------------------------------------- translated programs
def image_listing(fe, vgprs=0):
    """The hole image of a translated program as ONE listing, laid out as fx_xlate.cpp planXlate does:
    [steady fast][last fast][steady exact][last exact][run-once], each on a 64-byte boundary (s_nop filler).
    Returns (listing, bytes): the encoder's byte count, which the re-assembled listing must reproduce."""
    parts, at = [], 0
    for s in (0, 2, 1, 3, 4):
        code, listing = fe.translate(vgprs, s)
        if not code:
            continue   # no fast streams (a non-finite uniform operand), no run-once code
        parts.append(listing.strip())
        at += len(code)
        pad = (-at) % 64
        parts += ["s_nop 0"] * (pad // 4)
        at += pad
    return "\n".join(parts), at

def stream_entries(ins):
    """entry points of a hole image and the index mode they are reached with: an instruction nothing in the image falls or
    branches into is entered from the template (stream heads, cold stubs, the run-once code: index mode off); the
    instruction after an s_setpc_b64 is a handler's return point (whatever the handler left)."""
    by_addr = {i.addr: i for i in ins}
    has_pred = set()
    for k, i in enumerate(ins):
        if is_branch(i) and branch_target(i) in by_addr:
            has_pred.add(by_addr[branch_target(i)].index)
        if not ends_flow(i) and k + 1 < len(ins):
            has_pred.add(k + 1)
    out = {}
    for k, i in enumerate(ins):
        if k > 0 and ins[k - 1].mnem == "s_setpc_b64" and ins[k - 1].ops != ["s[34:35]"]:   # (s[34:35]: the way out, to the template's epilogue)
            out[i.addr] = UNKNOWN
        elif k not in has_pred:
            out[i.addr] = OFF
    return out


def staged_image_listing(fe, stages, vgprs=0):
    """The hole image of a program cut into pipeline stages (fx_xlate.cpp buildStagedImage): per stage
    [steady fast][last fast][steady exact][last exact], then the shared run-once code.  Returns (listing, bytes, stages,
    info, heads): heads[k] = byte offsets of stage k's four stream heads in the image (in the order fast, last fast, exact, last exact)."""
    parts, at, heads = [], 0, []
    code, listing, k, info = fe.translate_staged(stages, 0, 0, vgprs)
    if k < 2:
        return None
    for st in range(k):
        mine = []
        for s in (0, 2, 1, 3):
            code, listing, _, _ = fe.translate_staged(stages, st, s, vgprs)
            if not code:
                continue
            mine.append(at)
            parts.append(listing.strip())
            at += len(code)
            pad = (-at) % 64
            parts += ["s_nop 0"] * (pad // 4)
            at += pad
        heads.append(mine)
    code, listing, _, _ = fe.translate_staged(stages, 0, 4, vgprs)
    if code:
        parts.append(listing.strip())
        at += len(code)
        pad = (-at) % 64
        parts += ["s_nop 0"] * (pad // 4)
        at += pad
    return "\n".join(parts), at, k, info, heads


def barrier_counts(ins, head, heads, exits=("s[34:35]",)):
    """Barriers a wavefront executes on every path from the stream head `head` to the next head of its stage (the loop
    branch, the hand-over to the last-sample stream, a change to the exact stream and on to ITS loop branch) or to the way
    out (s_setpc_b64 s[34:35]).  Returns (set of counts at the next head, set of counts at the way out)."""
    by_addr = {i.addr: i for i in ins}
    start = by_addr[head].index
    seen, work = set(), [(start, 0, True)]
    at_head, at_exit = set(), set()
    while work:
        k, n, first = work.pop()
        i = ins[k]
        if i.addr in heads and not first:
            at_head.add(n)
            continue
        if (k, n) in seen:
            continue
        seen.add((k, n))
        if n > 64:
            raise RuntimeError("barrier count runs away")
        n2 = n + (1 if i.mnem == "s_barrier" else 0)
        if i.mnem == "s_setpc_b64":
            if i.ops and i.ops[0] in exits:
                at_exit.add(n2)
                continue
            if k + 1 < len(ins):
                work.append((k + 1, n2, False))   # a handler call: comes back behind it (handlers hold no barrier)
            continue
        if i.mnem == "s_endpgm":
            continue
        if is_branch(i):
            t = branch_target(i)
            if t in by_addr:
                work.append((by_addr[t].index, n2, False))
            if i.mnem == "s_branch":
                continue
        if k + 1 < len(ins):
            work.append((k + 1, n2, False))
    return at_head, at_exit
