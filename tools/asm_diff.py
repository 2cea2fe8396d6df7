"""Compare two `make asm` outputs kernel by kernel (labels and comments normalised): SAME / DIFF / NEW per kernel.
usage: python tools/asm_diff.py old.s new.s"""
import re, subprocess, sys

def funcs(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
        body = re.sub(r';.*', '', m.group(2))
        body = re.sub(r'\.LBB\d+_\d+', 'LBB', body)
        out[m.group(1)] = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith('.')]
    return out

def dem(n):
    d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    m = re.search(r'(k_\w+(<[^>]*>)?)', d)
    return m.group(1) if m else d[:70]

a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
print(len(a), 'kernels before,', len(b), 'after')
for n in b:
    if n in a:
        va = sum(1 for l in a[n] if l.startswith('v_')); vb = sum(1 for l in b[n] if l.startswith('v_'))
        print('SAME' if a[n] == b[n] else 'DIFF', f'{len(a[n]):6d} {len(b[n]):6d}  valu {va:5d} {vb:5d} ', dem(n))
    else:
        print('NEW ', f'{"":6s} {len(b[n]):6d}  valu {"":5s} {sum(1 for l in b[n] if l.startswith("v_")):5d} ', dem(n))
for n in a:
    if n not in b: print('GONE', dem(n))
