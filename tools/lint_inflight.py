#!/usr/bin/env python3
"""Static check of the compiled kernels (kokoro-align_amd/csrc/build/*.s from `make -C kokoro-align_amd/csrc asm`).

The forward / backtrace kernels issue their prefetch loads from inline asm and release the destination
registers with counted s_waitcnt statements.  hipcc does not know those registers are in flight: if its
register allocation makes it COPY one of them (v_mov from a register that an asm load writes) before the load
has landed - at a loop back-edge when it rotates registers, at a merge, ahead of a multi-register release - the
copy reads garbage.  That produced wrong paths under memory pressure and memory faults during development
(DESIGN.md section 7).

The check walks each kernel's instructions in program order with the hardware's rule (vmcnt is an in-order
counter: after `s_waitcnt vmcnt(N)` everything but the N youngest vector-memory operations has completed) and
reports every v_mov / DPP mov whose SOURCE is a register written by an inline-asm load that is not yet known to
have landed.  Program order is not execution order across branches, so this is a lint, not a proof: it is
exact for the straight-line unrolled frame code and for the copies a compiler puts in front of a back-edge.

The same for LDS: the tile kernels' frame loops read LDS from inline asm (ds_read) and wait with counted lgkmcnt (LDS
instructions complete in order; scalar loads share the counter, which only makes a wait stricter).  ANY instruction that
reads a register such a read may still be writing is reported.

A third check: an SGPR written by v_readfirstlane / v_readlane and read by a vector-memory instruction less than five
wait states later (hipcc pads its own code, not asm statements).

A second check covers the other thing hipcc cannot see inside inline asm: a store of more than 64 bits keeps
reading its data registers for two wait states after it has issued (gfx940+), so the next two instruction slots
must not write them.  The compiler honours that for its own stores; behind an inline-asm store it happily
refills a staging register in the very next instruction (first seen as checkpoints whose first dword came from
the next group, only under load).

    python tools/lint_inflight.py [file.s ...]      (default: every build/*.s)  exit status 1 when something is reported
"""
import os
import re
import sys

KERNELS = ("forward_ck", "forward_w16", "backtrace_rc", "backtrace_w16", "forward_tp", "forward_ts", "chunk_map")
VMEM = re.compile(r"\s*(global_load|global_store|buffer_load|buffer_store|flat_load|flat_store|global_atomic)\w*\s+(.*)")
WAIT = re.compile(r"\s*s_waitcnt\s+(.*)")
MOV = re.compile(r"\s*v_mov_b32(?:_e32|_dpp|_e64)?\s+(v[0-9]+),\s*(v[0-9]+)\b")
MOV64 = re.compile(r"\s*v_mov_b64(?:_e32)?\s+v\[([0-9]+):([0-9]+)\],\s*v\[([0-9]+):([0-9]+)\]")


def regs(tok):
    tok = tok.strip()
    if tok.startswith("v["):
        a, b = map(int, tok[2:-1].split(":"))
        return [f"v{i}" for i in range(a, b + 1)]
    return [tok] if re.fullmatch(r"v[0-9]+", tok) else []


def check(path):
    txt = open(path).read()
    report, total = [], 0
    for k in re.split(r"\n(?=_ZN2ka\w+:)", txt):
        m = re.match(r"(_ZN2ka\w+):", k)
        if not m or not any(x in m.group(1) for x in KERNELS):
            continue
        issued = 0                 # vector-memory operations issued so far (program order)
        pending = {}               # register -> issue index of the inline-asm load that writes it
        in_asm = False
        found = []
        lds_issued, lds_pending = 0, {}
        for ln in k.split("\n"):
            if "ASMSTART" in ln:
                in_asm = True
                continue
            if "ASMEND" in ln:
                in_asm = False
                continue
            md = re.match(r"\s*(ds_\w+)\s+(.*)", ln)
            if md:
                lds_issued += 1
                ops = [o.strip() for o in re.split(r",\s*", md.group(2).split(" offset")[0])]
                is_read = md.group(1).startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append"))
                for o in (ops[1:] if is_read else ops):
                    for r in regs(o):
                        if r in lds_pending:
                            found.append("LDS read in flight: " + ln.strip())
                if is_read:
                    for r in regs(ops[0]):
                        if in_asm and md.group(1).startswith("ds_read"):
                            lds_pending[r] = lds_issued
                        else:
                            lds_pending.pop(r, None)
                continue
            if lds_pending:
                mwl = WAIT.match(ln)
                if mwl:
                    ml = re.search(r"lgkmcnt\((\d+)\)", mwl.group(1))
                    if ml:
                        landed = lds_issued - int(ml.group(1))
                        for r in [r for r, i in lds_pending.items() if i <= landed]:
                            del lds_pending[r]
                elif re.match(r"\s*(v_|global_|buffer_|flat_)\w+\s", ln):
                    body = ln.split(";")[0]
                    parts = body.split(None, 1)
                    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
                    is_store = parts[0].startswith(("global_store", "buffer_store", "flat_store"))
                    for o in (ops if is_store else ops[1:]):
                        for r in regs(o.split()[0] if o else o):
                            if r in lds_pending:
                                found.append("LDS read in flight: " + ln.strip())
                    if not is_store and ops and not parts[0].startswith("v_cmp"):
                        for r in regs(ops[0]):
                            if r in lds_pending:
                                found.append("LDS read in flight (overwritten): " + ln.strip())
            mv = VMEM.match(ln)
            if mv:
                issued += 1
                if mv.group(1) == "global_load":
                    dst = mv.group(2).split(",")[0]
                    for r in regs(dst):
                        if in_asm:
                            pending[r] = issued
                        else:
                            pending.pop(r, None)      # a compiler-tracked load: hipcc waits for it itself
                continue
            mw = WAIT.match(ln)
            if mw:
                mc = re.search(r"vmcnt\((\d+)\)", mw.group(1))
                if mc:
                    landed = issued - int(mc.group(1))
                    for r in [r for r, i in pending.items() if i <= landed]:
                        del pending[r]
                continue
            mm = MOV.match(ln)
            srcs = []
            if mm:
                srcs = [mm.group(2)]
            else:
                m64 = MOV64.match(ln)
                if m64:
                    srcs = [f"v{i}" for i in range(int(m64.group(3)), int(m64.group(4)) + 1)]
            for r in srcs:
                if r in pending:
                    found.append(ln.strip())
            # any other instruction that WRITES a pending register ends its life as a load destination
            mo = re.match(r"\s*(v_\w+|ds_\w+)\s+(v\[?[0-9:]+\]?)", ln)
            if mo and not ln.lstrip().startswith(("v_cmp", "v_cmpx")):
                for r in regs(mo.group(2)):
                    if r in pending and not srcs:
                        pending.pop(r, None)
        # wide inline-asm stores: nothing may write their data registers within two wait states
        lines = [ln for ln in k.split("\n") if ln.strip() and not ln.lstrip().startswith((";", ".", "//")) and not ln.rstrip().endswith(":")]
        for i, ln in enumerate(lines):
            ms = re.match(r"\s*(global_store_dwordx[34]|buffer_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)", ln)
            if not ms:
                continue
            ops = [o.strip() for o in ms.group(2).split(",")]
            data = regs(ops[1]) if len(ops) > 1 else []
            waited, j = 0, i + 1
            while waited < 2 and j < len(lines):
                nx = lines[j]
                mn = re.match(r"\s*s_nop\s+(\d+)", nx)
                if mn:
                    waited += int(mn.group(1)) + 1
                else:
                    mo = re.match(r"\s*(v_\w+|ds_\w+|global_load\w*|buffer_load\w*)\s+(v\[?[0-9:]+\]?)", nx)
                    if mo and not nx.lstrip().startswith(("v_cmp", "v_cmpx", "ds_write", "ds_bpermute")) and set(regs(mo.group(2))) & set(data):
                        found.append(f"{ln.strip()}  ->  {nx.strip()}")
                    waited += 1
                j += 1
        # an SGPR written by the VECTOR unit (v_readfirstlane / v_readlane) must not be read by a vector-memory
        # instruction for five wait states; hipcc pads its own instructions, not those inside an asm statement
        # (seen: a load that used a pointer register's previous contents - the base without the offset)
        for i, ln in enumerate(lines):
            mr = re.match(r"\s*v_read(?:first)?lane_b32\s+(s[0-9]+)\b", ln)
            if not mr:
                continue
            sreg = int(mr.group(1)[1:])
            waited, j = 0, i + 1
            while waited < 5 and j < len(lines):
                nx = lines[j]
                mn = re.match(r"\s*s_nop\s+(\d+)", nx)
                if mn:
                    waited += int(mn.group(1)) + 1
                else:
                    if VMEM.match(nx):
                        used = set()
                        for a, b in re.findall(r"s\[([0-9]+):([0-9]+)\]", nx):
                            used |= set(range(int(a), int(b) + 1))
                        used |= {int(x) for x in re.findall(r"\bs([0-9]+)\b", nx)}
                        if sreg in used:
                            found.append(f"{ln.strip()}  ->  {nx.strip()}  ({waited} wait states)")
                    if re.match(r"\s*(s_|v_)\w+\s+s\[?%d\b" % sreg, nx) or re.match(r"\s*s_\w+\s+s\[%d:" % sreg, nx):
                        break      # the register is rewritten
                    waited += 1
                j += 1
        report.append((m.group(1), found))
        total += len(found)
    return report, total


def default_paths():
    import glob
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return sorted(glob.glob(os.path.join(here, "kokoro-align_amd", "csrc", "build", "*.s")))


def check_all(paths):
    report, total = [], 0
    for p in paths:
        r, t = check(p)
        report += r
        total += t
    return report, total


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sys.argv[1:] or default_paths()
    rep, total = check_all(paths)
    for name, found in rep:
        if found:
            print(f"{name}: {len(found)} hazards (copy of a register whose load may be in flight / write of a wide store's data), e.g. {found[:3]}")
    print(f"{len(rep)} kernels checked, {total} suspicious copies")
    sys.exit(1 if total else 0)
