#!/usr/bin/env python3
"""Scan a gfx950 ISA listing (hipcc --cuda-device-only -S) for the two hazards the compiler does not pad around an MFMA that sits
inside an `asm` statement: (1) a vector instruction writing a register the next MFMA reads as srcA / srcB, within 3 instructions;
(2) anything but an MFMA touching an MFMA's destination registers before two further MFMAs have issued (the compiler spilling or
copying an accumulator tile right behind the MFMA that writes it).  usage: scan_mfma_hazards.py kernel.s  ->  prints the hits"""
import re,sys
lines=[l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
def regs(tok):
    m=re.match(r'([va])\[(\d+):(\d+)\]',tok)
    if m: return [(m.group(1),i) for i in range(int(m.group(2)),int(m.group(3))+1)]
    m=re.match(r'([va])(\d+)$',tok)
    if m: return [(m.group(1),int(m.group(2)))]
    return []
hits=0
# (1) VALU write -> MFMA A/B read within 3 instrs; (2) MFMA D write -> non-MFMA read/write within N instrs
for idx,l in enumerate(lines):
    if l.startswith('v_mfma'):
        ops=[t.strip() for t in l.split(None,1)[1].split(',')]
        src=set(regs(ops[1])+regs(ops[2]))
        for back in range(1,4):
            p=lines[idx-back]
            if p.startswith('v_') and not p.startswith('v_mfma'):
                d=p.split(None,1)[1].split(',')[0].strip()
                if set(regs(d)) & src:
                    hits+=1
                    print('VALU->MFMA', back, '|', p, '|', l)
        dst=set(regs(ops[0]))
        n=0
        for fwd in range(1,12):
            if idx+fwd>=len(lines): break
            q=lines[idx+fwd]
            if q.startswith('v_mfma'):
                n+=8   # an intervening MFMA covers the latency
                if n>=16: break
                continue
            if q.startswith('s_nop'):
                continue
            toks=re.findall(r'[va]\[\d+:\d+\]|\b[va]\d+\b',q)
            used=set()
            for t in toks: used|=set(regs(t))
            if used & dst:
                print('MFMA-D->use', fwd, '|', l, '|', q); hits+=1
                break
print('hits',hits)
