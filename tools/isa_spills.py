"""Where a kernel spills: scratch loads / stores of the compiler's -S output listed in program order between phase markers (barriers, the first
global load / MFMA after a marker), with the count of vector instructions in between.  usage: python tools/isa_spills.py build/isa/k_*.s"""
import sys


def main():
    f = sys.argv[1]
    lines = open(f).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith('_Z17uph_solver_kernel') and ': ' in l][0]
    end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith('.amdhsa_kernel')][0]
    nb = 0
    ninst = 0
    nval = 0
    ev = []
    for i in range(start, end):
        l = lines[i].strip()
        if not l or (l.startswith(('.', ';', '//')) and not l.startswith('; UPHMARK')) or l.split()[0].endswith(':'):
            continue
        ninst += 1
        op = l.split()[0]
        if l.startswith('; UPHMARK'):
            ev.append((ninst, nval, 'MARK ' + l[10:]))
            continue
        if op.startswith('v_'):
            nval += 1
        if op == 's_barrier':
            nb += 1
            ev.append((ninst, nval, 'BAR %d' % nb))
        elif op.startswith('scratch_'):
            ev.append((ninst, nval, l.split(';')[0]))
        elif 'mfma' in op:
            if not ev or not ev[-1][2].startswith('MFMA'):
                ev.append((ninst, nval, 'MFMA'))
        elif op.startswith('global_load'):
            if not ev or not ev[-1][2].startswith('GLOAD'):
                ev.append((ninst, nval, 'GLOAD'))
        elif op.startswith('s_cbranch') or op.startswith('s_branch'):
            pass
    # compress runs of scratch ops
    out = []
    for e in ev:
        kind = e[2].split()[0] if not e[2].startswith('MARK') else e[2]
        if out and kind.startswith('scratch_') and out[-1][2] == kind and e[0] - out[-1][4] < 40:
            out[-1][3] += 1
            out[-1][4] = e[0]
        else:
            out.append([e[0], e[1], kind if kind.startswith('scratch_') else e[2], 1, e[0]])
    for o in out:
        print('%6d  valu %6d  %-22s x%d' % (o[0], o[1], o[2], o[3]))
    print('instructions', ninst, 'valu', nval)


main()
