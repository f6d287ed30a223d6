"""Register pressure of a stretch of a kernel's assembly (the compiler's -S output, tools/one_kernel.sh ... -DUPH_ISA_MARKS=1): live VGPRs per instruction between two
UPHMARK comments, from first-definition / last-use intervals of every vector register in that stretch (straight-line approximation: the sample code is one
predicated block), the peak, and WHAT is live at the peak -- each live register with the instruction that defined it.
usage: python tools/isa_pressure.py build/isa/k.s <from mark> <to mark> [occurrence]"""
import re
import sys

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


NO_DEF = ('global_store', 'scratch_store', 'ds_write', 'flat_store', 'buffer_store', 's_', 'v_cmp', 'v_readlane', 'v_readfirstlane', 'ds_bpermute_b32_noret')


def main():
    path, m0, m1 = sys.argv[1], sys.argv[2], sys.argv[3]
    occ = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith('_Z17uph_solver_kernel') and ': ' in l][0]
    marks = [i for i in range(start, len(lines)) if lines[i].strip().startswith('; UPHMARK ' + m0)]
    a = marks[occ]
    b = [i for i in range(a + 1, len(lines)) if lines[i].strip().startswith('; UPHMARK ' + m1)][0]
    ins = []
    for i in range(a, b):
        l = lines[i].strip()
        if l.startswith('; UPHMARK'):
            ins.append(('MARK', l[10:], [], []))
            continue
        if not l or l.startswith(('.', ';', '//')) or l.split()[0].endswith(':'):
            continue
        l = l.split(';')[0].strip()
        op = l.split()[0]
        args = l[len(op):].split(',')
        if op.startswith(NO_DEF) or op.startswith('s_'):
            d, u = [], regs(l[len(op):])
        else:
            d = regs(args[0]) if args else []
            u = regs(','.join(args[1:]))
            if op.startswith(('v_fmac', 'v_mac', 'v_mfma', 'v_writelane', 'v_cndmask')) or 'dpp' in l or 'sdwa' in l:
                u = u + d                # read-modify-write forms
        ins.append((op, l, d, u))
    n = len(ins)
    first, last, defat = {}, {}, {}
    for k, (op, l, d, u) in enumerate(ins):
        for r in u:
            if r not in first:
                first[r] = -1            # live into the stretch
            last[r] = k
        for r in d:
            if r not in first:
                first[r] = k
                defat[r] = k
            elif k > last.get(r, -2) and r in defat and last.get(r, -2) >= defat[r]:
                pass
            last[r] = max(last.get(r, k), k)
    # a register redefined after its last use starts a new value: split intervals at definitions that follow the previous value's last use
    events = []
    cur = {}
    for k, (op, l, d, u) in enumerate(ins):
        for r in u:
            if r in cur:
                cur[r][1] = k
            else:
                cur[r] = [-1, k, 'live-in']
        for r in d:
            if r in cur and r not in u:
                events.append((r, cur[r][0], cur[r][1], cur[r][2]))
                cur[r] = [k, k, l]
            elif r not in cur:
                cur[r] = [k, k, l]
            else:
                cur[r][1] = k
    for r, v in cur.items():
        events.append((r, v[0], v[1], v[2]))
    press = [0] * n
    for r, s, e, _ in events:
        for k in range(max(s, 0), e + 1):
            press[k] += 1
    peak = max(range(n), key=lambda k: press[k])
    print('%s .. %s: %d instructions, live VGPRs: peak %d at instruction %d (%s)' % (m0, m1, n, press[peak], peak, ins[peak][1][:70]))
    for k, (op, l, d, u) in enumerate(ins):
        if op == 'MARK':
            print('  at mark %-28s instruction %5d  live %3d' % (l, k, press[k]))
    step = max(1, n // 40)
    print('  profile (every %d instructions): %s' % (step, ' '.join(str(press[k]) for k in range(0, n, step))))
    live = sorted([(s, r, e, t) for r, s, e, t in events if max(s, 0) <= peak <= e])
    print('  live at the peak (%d registers; defined at / last used at / by):' % len(live))
    for s, r, e, t in live:
        print('    v%-3d %5d %5d  %s' % (r, s, e, t[:90]))


main()
