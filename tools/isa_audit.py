#!/usr/bin/env python
"""In-flight register audit of hand-scheduled loads (cdna_hip_programming.md 5.7 item 1).

hipcc treats the VGPR destination of an inline-asm load as written at `;;#ASMEND`: between the load and OUR `s_waitcnt` it may
read, copy, spill or reuse the register while the data is still in flight.  This tool replays the device assembly of one kernel
(`hipcc -S --cuda-device-only`) region by region: every inline-asm `global_load_*` enters a FIFO (VMEM returns in order), every
`s_waitcnt vmcnt(N)` retires all but the N youngest entries, and any instruction -- compiler-generated or ours -- that reads or
writes a register whose load is still in the FIFO is reported.  Inline-asm `ds_read_*` and `s_load_*` results are tracked the
same way on the LGKM counter (SGPRs included: hipcc spilled an in-flight scalar-load result once, see pna_fused_degree.hip).  Compiler-issued VMEM operations (loads, stores, LDS copies) are
entered too: they occupy counter slots, which only ever makes a counted wait stricter, so they are retired like the others.

The control-flow graph is explored exhaustively: both sides of every conditional branch, each (basic block, FIFO picture) state
once, so a loop is followed until its in-flight picture repeats.

    python tools/isa_audit.py <file.s> <kernel-name-substring> [--verbose]
Used by tests/test_build_resources.py.
"""
import re
import sys

_REG = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")
_SREG = re.compile(r"\bs(\d+)\b|s\[(\d+):(\d+)\]")


def _regs(rx, text, base=0):
    out = set()
    for m in rx.finditer(text):
        if m.group(1) is not None:
            out.add(base + int(m.group(1)))
        else:
            out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    return out


def vregs(text):
    return _regs(_REG, text)


def allregs(text):
    """VGPRs as 0.., SGPRs as 1000.. (one namespace for the in-flight sets)."""
    return _regs(_REG, text) | _regs(_SREG, text, 1000)


def kernel_lines(path, name):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        head = l.split(";")[0].strip()
        if start is None and head.endswith(":") and name in head and not l.startswith(("\t", ".")):
            start = i
        if start is not None and ".end_amdhsa_kernel" in l:
            return lines[start:i]
    raise SystemExit(f"kernel *{name}* not found in {path}")


def parse(lines):
    """-> list of (lineno, in_asm, mnemonic, operand text) for instructions; labels as (lineno, None, 'label', name)."""
    out, in_asm = [], False
    for i, raw in enumerate(lines):
        l = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l:
            continue
        if l.endswith(":") and not l.startswith("\t"):
            out.append((i, None, "label", l[:-1]))
            continue
        if l.startswith("."):
            continue
        parts = l.split(None, 1)
        out.append((i, in_asm, parts[0], parts[1] if len(parts) > 1 else ""))
    return out


def is_vmem(mn):
    return mn.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "scratch_load", "scratch_store", "flat_load", "flat_store",
                          "global_atomic"))


def _sgpr_result_is_dead(ins, labels, k, limit=4000):
    """True when the SGPR written by instruction k (a compiler-generated v_readfirstlane_b32 / v_readlane_b32) is overwritten before
    it is read on EVERY path from k (bounded walk of the control-flow graph).  hipcc materialises an undefined value of a
    structurizer flow block as `v_readfirstlane_b32 sN, v0` -- a lane read of WHATEVER sits in v0, possibly a register whose load is
    in flight -- and overwrites sN before anything looks at it: reading stale bits into a dead register is harmless."""
    dst = _regs(_SREG, ins[k][3].split(",")[0], 0)
    if len(dst) != 1:
        return False
    (d,) = dst
    stack, seen, steps = [k + 1], set(), 0
    while stack:
        j = stack.pop()
        while j < len(ins):
            if j in seen:
                break
            seen.add(j)
            steps += 1
            if steps > limit:
                return False
            _, _, mn, ops = ins[j]
            if mn == "label":
                j += 1
                continue
            if mn == "s_endpgm":
                break
            if mn == "s_branch":
                j = labels[ops.strip()]
                continue
            if mn.startswith("s_cbranch"):
                stack.append(labels[ops.strip()])
                j += 1
                continue
            parts = [x.strip() for x in ops.split(",")]
            writes = _regs(_SREG, parts[0], 0) if parts and (mn.startswith(("s_", "v_readlane", "v_readfirstlane", "v_cmp")) and not mn.startswith(("s_cmp", "s_waitcnt", "s_nop", "s_bitcmp", "s_setprio", "s_sleep", "s_barrier"))) else set()
            if "_co_" in mn and len(parts) > 1:
                writes = _regs(_SREG, parts[1], 0)
            reads = _regs(_SREG, ",".join(parts[1:]) if writes else ops, 0)
            if mn in ("s_cmp_eq_u32",) or mn.startswith(("s_cmp", "s_bitcmp")):
                reads = _regs(_SREG, ops, 0)
            if d in reads:
                return False
            if d in writes:
                break                                        # overwritten on this path: dead
            j += 1
    return True


def audit(lines, verbose=False):
    """Explores the kernel's control-flow graph (both sides of every conditional branch; a (block, FIFO) state is visited once, so
    loops converge as soon as their in-flight picture repeats).  Returns [(line, mnemonic, operands, registers)]."""
    ins = parse(lines)
    labels = {name: k for k, (_, a, mn, name) in enumerate(ins) if mn == "label"}
    problems, seen_prob, visited = [], set(), set()
    # (pc, VMEM fifo, LGKM fifo, known loop flags, vcc); fifo entries: sorted tuple of dst regs (empty for ops without a tracked dst).
    # Loop flags: hipcc lowers `break` out of a loop as `s_mov_b64 s[a:b], -1 / 0` on the two ways into a common block that then
    # tests `s_and_b64 vcc, exec, s[a:b]; s_cbranch_vccnz exit` -- without following that constant the audit walks "break, then the
    # loop header anyway", a path that does not exist (and on which loads requested for the next iteration are never waited for).
    stack = [(0, (), (), (), None)]
    steps = 0
    while stack:
        k, fifo, lgkm, flags, vcc = stack.pop()
        fifo, lgkm, flags = list(fifo), list(lgkm), dict(flags)
        while k < len(ins):
            ln, in_asm, mn, ops = ins[k]
            steps += 1
            if steps > 20_000_000:
                raise SystemExit("isa_audit: state explosion (more than 2e7 steps)")
            if mn == "label":
                key = (k, tuple(fifo), tuple(lgkm), tuple(sorted(flags.items())), vcc)
                if key in visited:
                    break
                visited.add(key)
                k += 1
                continue
            if mn == "s_endpgm":
                break
            if mn == "s_branch":
                k = labels[ops.strip()]
                continue
            if mn.startswith("s_cbranch"):
                taken = {"s_cbranch_vccnz": vcc, "s_cbranch_vccz": None if vcc is None else not vcc}.get(mn)
                if taken is True:
                    k = labels[ops.strip()]
                    continue
                if taken is None:
                    stack.append((labels[ops.strip()], tuple(fifo), tuple(lgkm), tuple(sorted(flags.items())), vcc))
                k += 1
                continue
            if not in_asm and (mn.startswith("s_") or mn.startswith("v_cmp") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane")
                               or "_co_" in mn):
                parts = [x.strip() for x in ops.split(",")]
                m = re.match(r"s_mov_b64$", mn) and re.match(r"s\[(\d+):(\d+)\]$", parts[0]) and parts[1] in ("0", "-1")
                if mn in ("s_and_b64", "s_andn2_b64") and parts[0] == "vcc" and parts[1] == "exec" and parts[2] in flags:
                    vcc = (flags[parts[2]] != 0) == (mn == "s_and_b64")          # (exec is never empty where a wavefront runs)
                else:
                    dst = parts[0] if not (("_co_" in mn) and len(parts) > 1) else parts[1]
                    if dst == "vcc" or dst.startswith("vcc"):
                        vcc = None
                    written = _regs(_SREG, dst, 0)
                    for key_ in [f for f in flags if _regs(_SREG, f, 0) & written]:
                        del flags[key_]
                    if m:
                        flags[parts[0]] = int(parts[1])
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    n = int(m.group(1))
                    if len(fifo) > n:
                        fifo = fifo[len(fifo) - n:] if n else []
                m = re.search(r"lgkmcnt\((\d+)\)", ops)
                if m:                # LDS reads return in order; scalar loads do not: only lgkmcnt(0) retires those
                    n = int(m.group(1))
                    if n == 0:
                        lgkm = []
                    elif not any(r and r[0] >= 1000 for r in lgkm) and len(lgkm) > n:
                        lgkm = lgkm[len(lgkm) - n:]
                k += 1
                continue
            inflight = set()
            for regs in fifo:
                inflight.update(regs)
            for regs in lgkm:
                inflight.update(regs)
            if in_asm and mn.startswith("s_load"):
                lgkm.append(tuple(sorted(_regs(_SREG, ops.split(",")[0], 1000))))
                k += 1
                continue
            if mn.startswith(";"):
                k += 1
                continue
            bad = allregs(ops) & inflight
            if bad and not in_asm and mn in ("v_readfirstlane_b32", "v_readlane_b32") and _sgpr_result_is_dead(ins, labels, k):
                bad = set()                                  # (an undefined value's lane read into a register nobody reads)
            if bad and (ln, tuple(sorted(bad))) not in seen_prob:
                seen_prob.add((ln, tuple(sorted(bad))))
                problems.append((ln, mn, ops, sorted(bad)))
            if in_asm and mn.startswith("ds_read"):
                lgkm.append(tuple(sorted(vregs(ops.split(",")[0]))))
            if is_vmem(mn):
                dst = ()
                if in_asm and "_load" in mn and "lds" not in mn:             # compiler loads: hipcc waits for those itself
                    dst = tuple(sorted(vregs(ops.split(",")[0])))
                fifo.append(dst)
                if len(fifo) > 63:
                    fifo = fifo[-63:]
            k += 1
    if verbose:
        print(f"{len(ins)} instructions, {steps} steps explored, {len(visited)} block states, {len(problems)} problems")
    return problems


def _valu_sgpr_writes(mn, ops):
    """SGPRs (1000..) a VALU instruction writes: lane reads, compares (e64 destination), carry-outs."""
    if not mn.startswith("v_"):
        return set()
    parts = [x.strip() for x in ops.split(",")]
    if mn.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
        return _regs(_SREG, parts[0], 1000)
    if "_co_" in mn or mn.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")):
        return _regs(_SREG, parts[1], 1000) if len(parts) > 1 else set()
    return set()


def sgpr_hazards(lines, wait_states=5):
    """"VALU writes SGPR -> VMEM reads that SGPR" needs 5 wait states on gfx9 / CDNA (ISA guide, manually inserted wait states).
    hipcc's hazard recognizer inserts the s_nop for its OWN memory instructions but does not look inside inline asm: an SGPR
    spill reloaded by v_readlane_b32 right in front of an asm `global_load ... s[a:b]` hands the load a stale base (round 3: the
    two-full-block tower instantiation of pna_fused_degree.hip faulted at address 0x1000 that way).  Reports every inline-asm
    VMEM instruction with fewer than `wait_states` instructions (s_nop N counts N + 1) between it and a VALU write of one of its
    SGPR operands, following the control-flow graph backwards.  Returns [(line, mnemonic, operands, writer line)]."""
    ins = parse(lines)
    labels = {name: k for k, (_, a, mn, name) in enumerate(ins) if mn == "label"}
    preds = {k: [] for k in range(len(ins))}
    for k, (_, _, mn, ops) in enumerate(ins):
        if mn in ("s_branch",) or mn.startswith("s_cbranch"):
            preds[labels[ops.strip()]].append(k)
        if k + 1 < len(ins) and mn not in ("s_branch", "s_endpgm"):
            preds[k + 1].append(k)
    out = []
    for k, (ln, in_asm, mn, ops) in enumerate(ins):
        if not (in_asm and is_vmem(mn)):
            continue
        uses = _regs(_SREG, ops, 1000)
        if not uses:
            continue
        # wait states already provided inside the same asm statement (an `s_nop` in front of the load) are instructions too
        stack, seen = [(p, wait_states) for p in preds[k]], set()
        while stack:
            p, need = stack.pop()
            if need <= 0 or (p, need) in seen:
                continue
            seen.add((p, need))
            _, _, pmn, pops = ins[p]
            if pmn == "label":
                stack.extend((q, need) for q in preds[p])
                continue
            if _valu_sgpr_writes(pmn, pops) & uses:
                out.append((ln, mn, ops, ins[p][0]))
                break
            cost = 1
            if pmn == "s_nop":
                cost = int(pops.strip() or 0) + 1
            stack.extend((q, need - cost) for q in preds[p])
    return out


_PK_F32 = re.compile(r"^v_pk_(add|mul|fma)_f32$")


def pk_src1_hi_selects(lines):
    """Packed-fp32 instructions whose LOW lane selects the HIGH half of src1 (op_sel[1] = 1).  On gfx950 such an instruction
    intermittently drops the update of its low-half result in lanes 48-63 while another wavefront of the SIMD issues MFMAs
    (tools/ubench/pk_opsel_mfma_repro.hip) -- hipcc emits the form when it vectorises scalar code over crossed
    register pairs.  Returns [(line, mnemonic, operands)]; tests/test_build_resources.py allows none in a source with MFMA kernels."""
    out = []
    for ln, _, mn, ops in parse(lines):
        if mn and _PK_F32.match(mn):
            m = re.search(r"op_sel:\[([01]),([01])(?:,([01]))?", ops)          # (src2 of an fma: not measured, treated alike)
            if m and (m.group(2) == "1" or m.group(3) == "1"):
                out.append((ln, mn, ops))
    return out


if __name__ == "__main__":
    kl = kernel_lines(sys.argv[1], sys.argv[2])
    probs = audit(kl, verbose=True)
    for ln, mn, ops, bad in probs:
        print(f"  line {ln}: {mn} {ops}   touches in-flight v{bad}")
    haz = sgpr_hazards(kl)
    for ln, mn, ops, w in haz:
        print(f"  line {ln}: {mn} {ops}   SGPR operand written by a VALU instruction at line {w}, fewer than 5 wait states before")
    pk = pk_src1_hi_selects(kl)
    for ln, mn, ops in pk:
        print(f"  line {ln}: {mn} {ops}   packed fp32 with the low lane reading src1's high half (wrong beside MFMA wavefronts)")
    sys.exit(1 if probs or haz or pk else 0)
