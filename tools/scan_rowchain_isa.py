#!/usr/bin/env python3
"""Static check of csrc/rowchain.hip's generated code (gfx950 assembly of one kernel on stdin or in argv[1]).

The row-chain kernels read their weight fragments with inline-asm ds_read_b128 and wait for them with hand-counted s_waitcnt
lgkmcnt(N).  The compiler regards an asm output as valid as soon as the statement ends: if it moves, copies or spills such a register
before the wait, it moves stale bytes (seen once: a v_accvgpr_write right behind the read, errors of 1e-4 that came and went with
unrelated edits).  This scan replays the LDS queue of the instruction stream - every ds_read enters it with its destination, every ds_write without one
(lgkmcnt counts both, in order), every
s_waitcnt lgkmcnt(N) retires all but the N youngest entries - and reports every instruction that touches the destination of a read
still in the queue.  tests/test_rowchain_isa.py runs it on every kernel of the file."""
import re, sys
lines = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith(';')]
def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
pending = []   # list of (regset, line_no, order)
bad = 0
for i, l in enumerate(lines):
    if l.startswith('ds_read'):                                  # every LDS operation takes a place in the in-order lgkm queue:
        dst = l.split()[1].rstrip(',')                           # reads with the registers they fill, writes with none
        pending.append((regs(dst), i))
        continue
    if l.startswith('ds_write'):
        pending.append((set(), i))
        continue
    m = re.match(r's_waitcnt.*lgkmcnt\((\d+)\)', l)
    if m:
        n = int(m.group(1))
        pending = pending[len(pending) - n:] if n and len(pending) > n else ([] if n == 0 else pending)
        continue
    if l.startswith(('s_', 'ds_write', 'buffer_', 'global_', 'scratch_')) and not l.startswith('scratch_store'):
        pass
    used = regs(l)
    for rs, ln in pending:
        if used & rs and not l.startswith('ds_read'):
            bad += 1
            if bad <= 12: print("HAZARD: line", i, l[:90], " reads regs of pending read at", ln, lines[ln][:60])
print("hazards:", bad, "of", len(lines), "instructions")
