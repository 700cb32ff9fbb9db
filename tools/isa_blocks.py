#!/usr/bin/env python3
"""Developer tool: instruction mix per basic block of named kernels in montecarlo_amd/csrc/amc_api.gfx950.s (make asm)."""
import re, collections, sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, 'montecarlo_amd/csrc/amc_api.gfx950.s')).read()
def kern(name):
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    return s[i:j].splitlines()
def blocks(lines):
    bl = []; cur = ['entry:']
    for l in lines:
        if re.match(r'^\.LBB\d+_\d+:', l): bl.append(cur); cur = [l]
        else:
            cur.append(l)
            t = l.strip()
            if t.startswith(('s_cbranch', 's_branch', 's_endpgm')):      # a basic block ends at every branch
                bl.append(cur); cur = ['  (fallthrough after ' + t.split()[0] + ' ' + (t.split()[1] if len(t.split()) > 1 else '') + '):']
    bl.append(cur); return bl
def ins(b): return [l.split()[0] for l in b if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
def cls(k):
    if k.startswith('v_readlane') or k.startswith('v_writelane'): return 'lane'
    if 'f64' in k: return 'f64'
    if k.startswith('v_mad_u64'): return 'mad64'
    if k.startswith('v_'): return 'v32'
    if k.startswith('s_'): return 'salu'
    if k.startswith('ds_'): return 'lds'
    if k.startswith(('buffer', 'global', 'flat')): return 'vmem'
    return 'other'
names = [a for a in sys.argv[1:] if not a.startswith('-')]
syms = re.findall(r'^(_ZN3amc[^:\s]*):', s, re.M)
for pat in names:
    for name in syms:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        if pat not in dn: continue
        print('==', dn)
        for b in blocks(kern(name)):
            I = ins(b)
            if len(I) < (1 if '-a' in sys.argv else 15): continue
            c = collections.Counter(cls(k) for k in I)
            print('  ', b[0].split(':')[0].ljust(44), len(I), dict(c), (b[-1].strip() if b[-1].strip().startswith(('s_cbranch','s_branch')) else ''))
            if '-v' in sys.argv:
                h = collections.Counter(I)
                print('      ', ', '.join(f'{k}:{v}' for k, v in h.most_common(30)))
