"""Audit of a gfx950 assembly listing (hipcc -save-temps): inside the named kernel, between an inline-assembly global_load into
vector registers and the next s_waitcnt vmcnt, no instruction OUTSIDE an inline-assembly block may touch those registers (hipcc
does not know the load is asynchronous: a copy or an arithmetic instruction it schedules there reads registers whose data has not
landed).  Linear scan in listing order, every wait clears the set -- conservative for straight-line issue -> wait sequences.
  python tools/isa_audit.py listing.s kernel_name_substring"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
    return out


def audit(text, kernel):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and l.split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    pending, inasm, bad, nloads = {}, False, [], 0
    for i in range(start, end):
        t = lines[i].strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if inasm and t.startswith("global_load") and "lds" not in t.split()[0]:
            for r in regs(t.split()[1].rstrip(",")):
                pending[r] = i
            nloads += 1
            continue
        if "s_waitcnt" in t and "vmcnt" in t:
            pending = {}
            continue
        if not inasm and re.match(r"(v_|global_|ds_|buffer_|flat_)", t):
            ops = t.split(None, 1)[1] if len(t.split(None, 1)) > 1 else ""
            for r in regs(ops):
                if r in pending:
                    bad.append((i + 1, t, r, pending[r] + 1))
    return nloads, bad


if __name__ == "__main__":
    n, bad = audit(open(sys.argv[1]).read(), sys.argv[2])
    for b in bad:
        print("line %d: %s   <- touches v%d, loaded at line %d and not yet waited for" % b)
    print("%d inline-assembly loads, %d violations" % (n, len(bad)))
    sys.exit(1 if bad or not n else 0)
