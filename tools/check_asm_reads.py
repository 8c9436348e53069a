#!/usr/bin/env python3
"""Static check of hand-scheduled LDS reads in a hipcc -S listing: between an inline-asm `ds_read_b64_tr_b16` and the next
`s_waitcnt lgkmcnt(N)` that retires it no instruction may read or overwrite the registers that read is going to deliver (hipcc does not
know the asm's result is asynchronous).  LDS reads retire in order: `lgkmcnt(N)` leaves the N youngest pending.  usage: check_asm_reads.py file.s"""
import re, sys
def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
bad = 0; pending = {}; order = []; kern = None; nread = 0
for ln, line in enumerate(open(sys.argv[1]), 1):
    l = line.split(';')[0].strip()
    if not l or l.startswith('.') : 
        if l.startswith('.amdhsa_kernel'): pending.clear(); order = []
        continue
    if l.endswith(':'):
        if not l.startswith('.L'): kern = l[:-1]; pending.clear(); order = []
        continue
    mw = re.search(r'lgkmcnt\((\d+)\)', l)
    if mw:
        keep = int(mw.group(1))
        for old in order[:max(0, len(order) - keep)]:
            for r in old:
                pending.pop(r, None)
        order = order[max(0, len(order) - keep):]
        continue
    parts = l.split(None, 1)
    op = parts[0]; ops = parts[1].split(',') if len(parts) > 1 else []
    if op == 'ds_read_b64_tr_b16':
        dst = regs(ops[0]); src = regs(ops[1])
        hit = src & set(pending)
        if hit: print("%s:%d address register pending: %s" % (kern, ln, l)); bad += 1
        for r in dst: pending[r] = ln
        order.append(sorted(dst))
        nread += 1
        continue
    if not pending: continue
    touched = set()
    for o in ops: touched |= regs(o)
    hit = touched & set(pending)
    if hit:
        print("%s:%d touches v%s pending since line %d: %s" % (kern[:40], ln, sorted(hit), min(pending[r] for r in hit), l)); bad += 1
print("checked %d tr-reads, %d violations" % (nread, bad))
sys.exit(1 if bad else 0)
