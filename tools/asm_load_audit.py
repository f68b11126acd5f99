#!/usr/bin/env python3
"""Audit of inline-asm register loads in the compiled kernels (run on the .s files of `hipcc -S --cuda-device-only`).

An inline-asm `global_load_* vDST, ...` is invisible to the compiler's wait-count pass: until an `s_waitcnt vmcnt(N)` that covers it has
executed, NOTHING may read or write vDST - yet the register allocator, which takes an asm output for a value that exists, is free to put
a copy (v_mov) or a spill of it in between.  It did: conv_halo's CF epilogue (round 5) - the bias loads' registers were copied in front
of the inline-asm wait that had them as tied "+v" operands, and whenever the loads missed in L2 a tile started from garbage.  This tool
walks every kernel's control-flow graph (forward data flow over basic blocks; state = the asm loads possibly in flight with the number of
vector-memory operations issued behind each) and prints every instruction that touches the destination of a load that may still be in
flight.  No finding = every asm load is first touched behind a wait that covers it.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -I include -I diffusion-nlc_amd/csrc -S --cuda-device-only -o x.s diffusion-nlc_amd/csrc/conv_halo.hip
    python tools/asm_load_audit.py x.s
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
CAP = 24          # younger-operation counts saturate here (no counted wait in the kernels allows more)


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return frozenset(out)


def functions(lines):
    start, name = None, None
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and ln.split(";")[0].rstrip().endswith(":"):
            start, name = i, ln.split(":")[0]
        elif ln.startswith(".Lfunc_end") and start is not None:
            yield name, start, i
            start = None


def audit_function(path, name, lines, lo, hi, report):
    # basic blocks
    blocks, cur, label_of = [], None, {}
    in_asm = False
    for i in range(lo + 1, hi):
        raw = lines[i]
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        code = raw.split(";")[0].strip()
        if not code:
            continue
        if code.startswith(".LBB") and code.endswith(":"):
            cur = {"label": code[:-1], "ins": [], "succ": [], "fall": True}
            label_of[code[:-1]] = len(blocks)
            blocks.append(cur)
            continue
        if code.startswith("."):
            continue
        if cur is None:
            cur = {"label": None, "ins": [], "succ": [], "fall": True}
            blocks.append(cur)
        cur["ins"].append((i, code, in_asm))
        op = code.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["succ"].append(code.split()[-1])
            nxt = {"label": None, "ins": [], "succ": [], "fall": True}
            if op == "s_branch":
                cur["fall"] = False
            blocks.append(nxt)
            cur = nxt
        elif op in ("s_endpgm", "s_setpc_b64"):
            cur["fall"] = False
            nxt = {"label": None, "ins": [], "succ": [], "fall": True}
            blocks.append(nxt)
            cur = nxt
    nb = len(blocks)
    edges = []
    for k, b in enumerate(blocks):
        e = [label_of[t] for t in b["succ"] if t in label_of]
        if b["fall"] and k + 1 < nb:
            e.append(k + 1)
        edges.append(e)
    state_in = [set() for _ in range(nb)]
    work = [0]
    found = set()
    seen_in = [None] * nb
    while work:
        k = work.pop()
        st = set(state_in[k])
        if seen_in[k] is not None and seen_in[k] == st:
            continue
        seen_in[k] = set(st)
        for (i, code, in_asm) in blocks[k]["ins"]:
            op = code.split()[0]
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                st = {q for q in st if q[2] < n}      # a load with >= n younger operations behind it is complete
                continue
            rest = code.split(None, 1)[1] if len(code.split(None, 1)) > 1 else ""
            touched = regs(rest)
            for q in st:
                if touched & q[1]:
                    found.add((i, q[0], min(q[1]), max(q[1]), code))
            if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
                st = {(a, b, min(c + 1, CAP)) for (a, b, c) in st}
                if in_asm and re.match(r"global_load_dword(x\d)?\s+v", code) and "lds" not in op:
                    st.add((i, regs(code.split(",")[0]), 0))
        for t in edges[k]:
            if not st <= state_in[t]:
                state_in[t] |= st
                work.append(t)
            elif seen_in[t] is None:
                work.append(t)
    for (i, at, r0, r1, code) in sorted(found):
        report.append(f"{path}:{i + 1}: [{name}] touches v{r0}..v{r1} (asm load at line {at + 1}) while it may be in flight: {code}")
    return len(found)


def audit(path):
    lines = open(path).read().split("\n")
    total, report = 0, []
    for name, lo, hi in functions(lines):
        total += audit_function(path, name, lines, lo, hi, report)
    for r in report[:40]:
        print(r)
    if len(report) > 40:
        print(f"... and {len(report) - 40} more")
    return total


if __name__ == "__main__":
    total = 0
    for pth in sys.argv[1:]:
        total += audit(pth)
    print(f"{total} touches of in-flight asm-load registers")
    sys.exit(1 if total else 0)
