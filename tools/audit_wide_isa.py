#!/usr/bin/env python3
"""Audit of the wide stack kernels' ISA (usage: audit_wide_isa.py file.s).  The accumulators live in fixed registers (a0..a255, v192..) that the
compiler tracks as pinned 32-register values (mshgnn_wide.hip, WRegs): correctness is the compiler's, but every instruction IT generates on those
registers (tuple copies at loop back edges, evacuations under register pressure) is time the MAC engine's design did not budget -- this lists them
per kernel, with code size and scratch.  Advisory: exit code 0."""
import re, sys
s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_ZN12_GLOBAL__N_1\w+:)', s)
bad_total = 0
for f in funcs[1:]:
    name = f.split(':')[0]
    if 'k_wide' not in name and 'k_eng' not in name: continue
    m = re.search(r'ILi(\d+)ELi(\d+)E', name)
    nh, ns = int(m.group(1)), int(m.group(2))      # geometry (halves per tile) and accumulator slots: slots 16.. live in v[VACC ..], VACC = 192 (wide) / 112 (slab2)
    inasm = False; bad = []
    for i, l in enumerate(f.split('\n')):
        if ';;#ASMSTART' in l: inasm = True; continue
        if ';;#ASMEND' in l: inasm = False; continue
        if inasm or l.strip().startswith(';') or l.strip().startswith('.'): continue
        if re.search(r'\ba\[?\d+', l) or 'accvgpr' in l: bad.append((i, l.strip()))
        regs = [int(x) for x in re.findall(r'\bv(\d+)\b', l)] + [int(y) for x in re.findall(r'v\[(\d+):(\d+)\]', l) for y in x]
        vacc = 192 if nh == 2 else 112
        if any(vacc <= r < vacc + 16 * nh * ((max(0, ns - 16) + 1) // 2) for r in regs): bad.append((i, l.strip()))
    m = re.search(r'; ScratchSize: (\d+)', f); c = re.search(r'; codeLenInByte = (\d+)', f)
    print(f"{name[19:56]:38s} code {c.group(1) if c else '?':>7s} B  scratch {m.group(1) if m else '?':>4s} B  compiler-generated instructions on accumulator registers: {len(bad)}", bad[:3])
    bad_total += len(bad)
sys.exit(0)
