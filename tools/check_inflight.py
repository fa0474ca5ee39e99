#!/usr/bin/env python3
"""tools/check_inflight.py <object.o> <kernel-name-regex> — static check of kernels that issue loads from inline asm and wait for them by
hand (lanczos3_strip2, lanczos3_strip<T>): between a `global_load_dwordx4` into a register quad and the hand-written `s_waitcnt vmcnt(N)`
that precedes the quad's first use, NO instruction may read or write the quad — the compiler does not know the registers are still being
written and is free to copy or reuse them (it did, in two earlier versions of lanczos3_strip<T>: garbage pixels and a memory fault).

Method (conservative, per kernel): the prefetch quads are the destinations of the x4 loads that are re-loaded somewhere (a quad loaded once is
an ordinary compiler-managed load).  Walking the code in program order, a quad is "in flight" from a load into it until the next
`s_waitcnt vmcnt(n)` with n smaller than the number of prefetch quads; while in flight, any other instruction naming one of its registers is a
violation.  The code is walked in program order; the kernels' loops are entered with every prefetch quad in flight (the prologue's loads) and come
back to their top in the same state, so one walk covers the back edge as well.  Prints the violations and exits 1 if there are any."""
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import isa_cost  # noqa: E402

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(code):
    out = set()
    for m in REG.finditer(code):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out |= set(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(text, pattern):
    bad, seen = [], 0
    # split the disassembly at symbol lines ("0000000000001234 <name>:")
    parts, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            parts[cur] = []
        elif cur is not None and line.strip():
            parts[cur].append(line.split("//")[0].rstrip())
    for name, body in parts.items():
        if not re.search(pattern, name):
            continue
        loads = [re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l) for l in body]
        count = {}
        for m2 in loads:
            if m2:
                q = (int(m2.group(1)), int(m2.group(2)))
                count[q] = count.get(q, 0) + 1
        quads = {q for q, n in count.items() if n > 1}
        if not quads:
            continue
        seen += 1
        inflight = set()
        for _pass in range(1):                      # (program order; see the docstring for the loop's back edge)
            for n, l in enumerate(body):
                m2 = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
                if m2 and (int(m2.group(1)), int(m2.group(2))) in quads:
                    q = (int(m2.group(1)), int(m2.group(2)))
                    addr = regs_of(l.split(",", 1)[1])
                    for f in inflight:
                        if addr & set(range(f[0], f[1] + 1)):
                            bad.append((name, n, l.strip(), f"address uses in-flight quad v[{f[0]}:{f[1]}]"))
                    if q in inflight:
                        bad.append((name, n, l.strip(), "reloaded while in flight"))
                    inflight.add(q)
                    continue
                w = re.search(r"s_waitcnt vmcnt\((\d+)\)", l)
                if w:
                    if int(w.group(1)) < len(quads):
                        # in order among loads: everything but the youngest `n` has arrived; the walk does not track ages, and the
                        # kernels only ever touch the oldest quad after such a wait, so: the quads touched before the next load are free
                        inflight_after_wait = set(inflight)
                        pending_release = True
                    continue
                used = regs_of(l)
                for f in list(inflight):
                    if used & set(range(f[0], f[1] + 1)):
                        if locals().get("pending_release") and f in inflight_after_wait:
                            inflight.discard(f)          # first touch after a partial wait: this is the quad that was awaited
                            pending_release = False
                        else:
                            bad.append((name, n, l.strip(), f"touches in-flight quad v[{f[0]}:{f[1]}]"))
    return seen, bad


if __name__ == "__main__":
    text = isa_cost.disassemble(sys.argv[1])
    seen, bad = check(text, sys.argv[2])
    print(f"{seen} kernel(s) with hand-awaited loads checked, {len(bad)} violation(s)")
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad or not seen else 0)
