#!/usr/bin/env python3
"""Developer tool: instruction mix of the sweep kernel's inner loop with measured issue costs."""
import re, collections, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args=[a for a in sys.argv[1:] if not a.startswith('-')]
name = args[0] if args else '_ZN3amc12sweep_kernelILi0ELb0ELb0ELb0ELb1ELb0EEEvNS_9SweepArgsE'
subprocess.run(['/opt/rocm/bin/hipcc','-O3','-std=c++17','--offload-arch=gfx950','-ffp-contract=off','-fno-fast-math','-S','--cuda-device-only',
                os.path.join(ROOT,'montecarlo_amd/csrc/amc_api.hip'),'-o','/tmp/amc.s'],check=True,stderr=subprocess.DEVNULL)
s=open('/tmp/amc.s').read()
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
lines=s[i:j].splitlines()
blocks=[];cur=[]
for l in lines:
    if re.match(r'^\.LBB\d+_\d+:',l): blocks.append(cur);cur=[l]
    else: cur.append(l)
blocks.append(cur)
cost={'v_fma_f64':5.5,'v_mul_f64':5.5,'v_add_f64':5.5,'v_fmac_f64_e32':5.5,'v_mad_u64_u32':5.1,'v_mul_hi_u32':4.3,'v_mul_lo_u32':4.3,'v_rcp_f64_e32':17.7,'v_rsq_f64_e32':17}
def w(k):
    if k in cost: return cost[k]
    if not k.startswith('v_'): return 0
    return 2.7 if any(t in k for t in ('b32','u32','i32','_f32')) and 'f64' not in k else 4.5
print("blocks (label, n_instr, est units):")
for b in blocks:
    ins=[l.split()[0] for l in b if l.startswith('\t') and not l.strip().startswith(('.',';'))]
    if len(ins)>=20: print("  ",b[0].split(':')[0], len(ins), round(sum(w(k) for k in ins)))
tot=collections.Counter()
big=max(blocks,key=len)
for l in big:
    if l.startswith('\t') and not l.strip().startswith(('.',';')): tot[l.split()[0]]+=1
print("histogram of the largest block", big[0].split(':')[0])
if '-v' in sys.argv:
    for k,v in tot.most_common(40): print(f"{v:4d} {k:28s} {w(k)*v:7.1f}")
md=s[s.index('amdhsa.kernels'):]
for b in md.split('  - .agpr_count')[1:]:
    n=re.search(r'\.name:\s+(\S+)',b).group(1)
    if n==name:
        g=lambda k: re.search(r'\.'+k+r':\s+(\S+)',b).group(1)
        print('vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'sgpr_spill',g('sgpr_spill_count'),'vgpr_spill',g('vgpr_spill_count'),'lds',g('group_segment_fixed_size'))
