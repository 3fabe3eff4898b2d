#!/bin/bash
# instruction histogram of the tile loop of the cached FVP instance (the largest backward branch span)
cd /tmp && mkdir -p exp && cd exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S -o mjx.s /root/repo/mjrl_amd/csrc/mjx.hip 2>/dev/null
awk '/^_ZN3mjx7k_fusedILi64ELi64ELi1ELi8ELi1ELb0ELi20ELb1EEEvNS_9FusedArgsE:/,/s_endpgm/' mjx.s > fvp.s
python3 - <<'PY'
import re, collections
lines=open('fvp.s').read().split('\n')
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
best=(0,0,0)
for i,l in enumerate(lines):
    m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i and i-labels[m.group(1)]>best[0]: best=(i-labels[m.group(1)],labels[m.group(1)],i)
body=lines[best[1]:best[2]+1]
c=collections.Counter()
for l in body:
    l=l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
    c[l.split()[0]]+=1
print("loop body: %d instructions"%sum(c.values()))
print(", ".join("%d %s"%(v,k) for k,v in c.most_common(40)))
PY
grep -E "vgpr_count|vgpr_spill|scratch" fvp.s | head -5
