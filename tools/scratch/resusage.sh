#!/bin/bash
# tools/scratch/resusage.sh [file.hip]: VGPRs / scratch / spills per kernel of a source file
f=${1:-smallhardface_amd/csrc/conv_f16x3.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ismallhardface_amd/csrc -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/resusage.o 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass',l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'): cur={'name':t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
seen=set()
for r in rows:
    if r['name'] in seen: continue
    seen.add(r['name'])
    print('%-90s vgpr %4s scratch %4s sspill %3s vspill %3s' % (r['name'][:90], r.get('VGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('SGPRs Spill'), r.get('VGPRs Spill')))
"
