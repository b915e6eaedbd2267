#!/usr/bin/env python3
"""Compact view of a kernel's instruction stream from hipcc -S output: runs of the same instruction class collapsed
(MFMA x16, DSR x8, GLDS x2, WAIT vmcnt(6), BARRIER ...), to check by eye that a hand-scheduled loop came out as written --
fragment reads and LDS-DMA in front of the barrier, matrix instructions between barriers, no stray vmcnt(0).
    tools/isa_summary.py kernel.s <mangled-name-substring> [context-before]"""
import re
import sys


def cls(l):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        return None
    op = l.split()[0]
    if op.startswith('v_mfma'):
        return 'MFMA'
    if op.startswith('ds_read'):
        return 'DSR'
    if op.startswith('ds_write'):
        return 'DSW'
    if op.startswith('buffer_load') and ' lds' in l:
        return 'GLDS'
    if op.startswith(('buffer_load', 'global_load')):
        return 'VLOAD'
    if op.startswith(('buffer_store', 'global_store')):
        return 'VSTORE'
    if op == 's_barrier':
        return 'BARRIER'
    if op == 's_waitcnt':
        return 'WAIT ' + ' '.join(l.split()[1:])
    if op == 's_setprio':
        return 'PRIO ' + l.split()[1]
    if op.startswith('s_cbranch') or op == 's_branch':
        return 'BR ' + l.split()[-1]
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        return 'salu'
    return op


def main():
    s = open(sys.argv[1]).read()
    m = re.search(r'^(\S*%s\S*):' % re.escape(sys.argv[2]), s, flags=re.M)
    start = m.start()
    end = s.index('.Lfunc_end', start)
    before = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    out, prev, cnt = [], None, 0
    for l in s[start:end].splitlines():
        if re.match(r'^\.LBB\d+_\d+:', l.strip()):
            if prev:
                out.append((prev, cnt))
            out.append((l.strip(), 1))
            prev, cnt = None, 0
            continue
        c = cls(l)
        if c is None:
            continue
        if c == prev:
            cnt += 1
        else:
            if prev:
                out.append((prev, cnt))
            prev, cnt = c, 1
    if prev:
        out.append((prev, cnt))
    txt = ["%s x%d" % (a, b) if b > 1 else a for a, b in out]
    idx = [i for i, x in enumerate(txt) if x.startswith('MFMA')]
    print('%s: %d items, MFMA runs %d' % (m.group(1), len(txt), len(idx)))
    print('\n'.join(txt[max(0, idx[0] - before):idx[-1] + 14]))


if __name__ == '__main__':
    main()
