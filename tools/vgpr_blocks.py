"""Where a kernel's VGPR pressure peaks: per basic block of one kernel in a `make asm` output, the highest VGPR index referenced and
what kind of memory / division instructions the block holds.   vgpr_blocks.py file.s <mangled-name-substring> [threshold]"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 96
m = re.search(r'^(_Z[^\n:]*' + re.escape(key) + r'[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M)
print(m.group(1))
blocks, cur, name = [], [], 'entry'
for l in m.group(2).split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.split(':')[0]; cur = []
    else:
        cur.append(l)
blocks.append((name, cur))
for n, ls in blocks:
    ins = [l.split(';')[0] for l in ls if l.strip() and not l.strip().startswith(('.', ';'))]
    mx = -1
    for l in ins:
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', l): mx = max(mx, int(b))
        for a in re.findall(r'\bv(\d+)\b', l): mx = max(mx, int(a))
    if mx >= thr:
        kinds = {}
        for i in ins:
            op = i.split()[0]
            if op.startswith(('global_', 'ds_', 'v_div', 'v_rcp', 'scratch_', 'v_pk_', 'v_readlane', 'v_writelane')): kinds[op] = kinds.get(op, 0) + 1
        print(f"{n:12s} {len(ins):5d} instr  max v{mx:3d}  " + ' '.join(f"{k}x{v}" for k, v in sorted(kinds.items())))
