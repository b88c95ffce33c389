#!/usr/bin/env python3
"""Static VALU / SALU / memory instruction counts per basic block of ONE kernel in a gfx950 assembly listing, with the branch targets of
each block -- the census behind docs/EXPERIMENTS.md R5.3 (where do the instructions of k_solve_gcf_g go: set-up, loop body, tail).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIBS_WITH_F32 -DIBS_P=32 -DIBS_M=16 -S --cuda-device-only -o /tmp/g.s ideal-ballooning-solver_amd/csrc/ibs_kernels_group.hip
   python tools/isa_regions.py /tmp/g.s _ZN3ibs13k_solve_gcf_gIdLi16ELi32EdEE      (a prefix of the mangled kernel name)"""
import re,sys
fn, kern = sys.argv[1], sys.argv[2]
lines=open(fn).read().split('\n')
start=next(i for i,l in enumerate(lines) if l.startswith(kern) and l.rstrip().endswith(':') or (l.startswith(kern) and ':' in l))
end=next(i for i in range(start,len(lines)) if 's_endpgm' in lines[i])
# basic blocks
blocks=[]; cur=('entry',[])
for l in lines[start+1:end+1]:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        blocks.append(cur); cur=(m.group(1),[])
    elif l.startswith('\t') and not l.startswith('\t.') and not l.strip().startswith(';'):
        cur[1].append(l.strip())
blocks.append(cur)
tot=0
for name,ins in blocks:
    v=sum(1 for x in ins if x.startswith('v_'))
    s=sum(1 for x in ins if x.startswith('s_'))
    mem=sum(1 for x in ins if x.startswith(('global_','ds_','buffer_','flat_','scratch_')))
    br=[x for x in ins if x.startswith('s_cbranch') or x.startswith('s_branch')]
    tot+=v
    print("%-12s valu %5d salu %4d mem %4d  %s"%(name,v,s,mem,' '.join(b.split()[-1]+('('+b.split()[0][2:]+')') for b in br)))
print("total static valu",tot)
