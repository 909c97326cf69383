"""Check compiled gfx950 assembly for the hazard of LDS reads issued from inline asm (`ds_read*` between
`;;#ASMSTART` / `;;#ASMEND`): the hardware writes their destination registers LATER, but to the compiler the values exist
when the asm ends — so it may copy them (stale data) or, when it thinks them dead, hand the registers to something else
(the late write then lands in, say, an address).  Rule enforced here: from such a read, along every path
of the control-flow graph up to the `s_waitcnt lgkmcnt(N)` that retires it (LDS operations retire in order: a wait for N
retires it once N younger LDS operations have been issued), no instruction may read or write its destination register.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s kernel.hip && python scripts/check_async_lds_reads.py k.s

Found with it (round 5): `gemm_tile128.hip`'s last K step fetched fragments nobody multiplies; hipcc reused their
registers for the epilogue's addresses while the reads were in flight (a fault at K = 64).  tests/test_isa_async_reads.py
runs it on the kernels that use the pattern.
"""
import re
import sys

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_ASYNC = re.compile(r"^\s*ds_read\w*\s+(v\[\d+:\d+\]|v\d+)\s*,")
_LABEL = re.compile(r"^([.\w$]+):")
_WAIT = re.compile(r"^\s*s_waitcnt\b")
_BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+([.\w$]+)")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def functions(lines):
    """[(name, [line, ...])] of the kernels / functions in an assembly listing (from a `name:` label to s_endpgm / .Lfunc_end)."""
    out, cur, name = [], None, None
    for ln in lines:
        m = _LABEL.match(ln)
        if m and not m.group(1).startswith(".L") and cur is None and not ln.startswith("\t"):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln.rstrip("\n"))
            if ln.startswith(".Lfunc_end"):
                out.append((name, cur))
                cur = None
    return out


def check_function(name, body):
    """Violations in one function: [(line number in the function, text, registers)]."""
    code = [(i, ln.split(";")[0].rstrip() if not ln.lstrip().startswith(";;#") else ln.strip()) for i, ln in enumerate(body)]
    labels = {}
    for i, ln in code:
        m = _LABEL.match(ln)
        if m:
            labels[m.group(1)] = i
    n = len(code)
    violations = []
    in_asm = False
    for i in range(n):
        text = code[i][1]
        if text.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if text.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = _ASYNC.match(text) if in_asm else None
        if not m:
            continue
        dests = _regs(m.group(1))
        # path state: (pc, LDS operations issued since this read) — LDS operations retire in order, so a wait for
        # lgkmcnt(N) retires the read once at least N younger LDS operations have been issued
        seen, work = set(), [(i + 1, 0)]
        while work:
            pc, younger = work.pop()
            while pc < n and (pc, younger) not in seen:
                seen.add((pc, younger))
                ins = code[pc][1].strip()
                if not ins or ins.startswith(";;#") or _LABEL.match(ins) or ins.startswith("."):
                    pc += 1
                    continue
                if _WAIT.match(ins):
                    w = re.search(r"lgkmcnt\((\d+)\)", ins)
                    if w and int(w.group(1)) <= younger:
                        break
                    pc += 1
                    continue
                hit = _regs(ins) & dests
                if hit:
                    violations.append((pc, ins, sorted(hit), i))
                    break
                if ins.startswith("ds_"):
                    younger = min(younger + 1, 64)
                if ins.startswith("s_endpgm"):
                    break
                b = _BRANCH.match(ins)
                if b:
                    if b.group(2) in labels:
                        work.append((labels[b.group(2)], younger))
                    if b.group(1) == "s_branch":
                        break
                pc += 1
    return violations


def check_file(path):
    with open(path) as f:
        lines = f.readlines()
    report, blocks = [], 0
    for name, body in functions(lines):
        in_asm, has = False, 0
        for ln in body:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and _ASYNC.match(ln):
                has += 1
        if not has:
            continue
        blocks += has
        for pc, ins, regs, start in check_function(name, body):
            report.append(f"{name}: line +{pc}: `{ins}` touches v{regs} while the asm reads issued at +{start} are in flight")
    return blocks, report


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        blocks, report = check_file(p)
        print(f"{p}: {blocks} asynchronous LDS reads from inline asm, {len(report)} violation(s)")
        for r in report:
            print("  " + r)
        bad += len(report)
    sys.exit(1 if bad else 0)
