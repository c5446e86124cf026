#!/usr/bin/env python3
"""Build-time guard for the event aligner's hand-awaited loads (Weak 9; nanopolish_amd/csrc/Makefile runs it on every build
of np_align_kernel.hip and FAILS the build on a violation).

np_align_kernel.hip issues some of its loads from inline asm and waits for them with inline `s_waitcnt vmcnt(n)` -- the back-track's
trace prefetch queue (NP_BT_DEPTH groups in flight) and the band loop's event prefetch.  The compiler does not know such a register
is in flight: if it spills a queue entry, or copies an asm output "right away", it reads the register before the data has landed and
the kernel walks garbage (seen once, at NP_BT_DEPTH = 12).  This script reads the assembly hipcc generates (-S --cuda-device-only) and
checks, for every np_event_align_kernel instantiation, that no compiler-generated instruction -- a copy, a spill store, anything --
reads or overwrites the destination registers of an inline-asm load while that load may still be in flight.  "In flight" is modelled
as the hardware counts it: every vector-memory load (the asm's and the compiler's own, spill reloads included) joins a queue in
program-text order, `s_waitcnt vmcnt(n)` -- hand-written or the compiler's -- retires all but the n youngest.  (Text order stands in
for execution order; an unconditional branch and the kernel's outermost loop header reset the model: what follows them is reached
from elsewhere.)  Spills of other values are
allowed: a reload only makes a hand-written wait stronger.
Usage: check_align_isa.py file.s  (exit status 1 and a message per violation)."""
import re
import sys


def regs(tok):
    """v5 -> {5}; v[4:7] -> {4,5,6,7}; anything else -> empty"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    L = open(sys.argv[1]).read().splitlines()
    bad = []
    names = [(i, l.split(":")[0]) for i, l in enumerate(L) if l.startswith("_Z") and "np_event_align_kernel" in l.split(":")[0]]
    if not names:
        bad.append("no np_event_align_kernel found in %s" % sys.argv[1])
    for start, name in names:
        end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
        body = L[start:end]
        in_asm = False
        queue = []              # loads in flight, oldest first: (set of asm destination registers or None for a compiler load, line)
        for off, l in enumerate(body):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True; continue
            if t.startswith(";;#ASMEND"):
                in_asm = False; continue
            if "Loop Header: Depth=1" in t:
                queue = []
            if not t or t.startswith(";") or t.startswith("."):
                continue
            op = t.split()[0]
            args = [a.strip() for a in t[len(op):].split(",")] if len(t) > len(op) else []
            is_load = op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load"))
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    queue = queue[len(queue) - int(m.group(1)):] if int(m.group(1)) < len(queue) else queue
                    if int(m.group(1)) == 0:
                        queue = []
                continue
            if op == "s_endpgm":
                break
            if op in ("s_branch", "s_setpc_b64"):
                queue = []          # the text that follows is reached from somewhere else: its queue state is unknown (taken as empty)
                continue
            if not in_asm:
                touched = set()
                for a in args:
                    touched |= regs(a.split()[0] if a else "")
                for dests, line in queue:
                    if dests and touched & dests:
                        bad.append("%s: line %d `%s` touches v%s while the inline-asm load of line %d may still be in flight"
                                   % (name, start + off + 1, t, sorted(touched & dests), line))
            if is_load and args:
                queue.append((regs(args[0]) if in_asm else None, start + off + 1))
    for b in bad:
        print("check_align_isa: " + b, file=sys.stderr)
    if bad:
        sys.exit(1)
    print("check_align_isa: %d kernel(s), no compiler access to a register an inline-asm load may still be writing" % len(names))


main()
