#!/bin/bash
# kernel resource summary of one csrc/*.hip file: name, VGPRs, scratch bytes, spills, LDS
f=$1
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'n':subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()[:110]}; rows.append(cur)
    for k,p in (('v',r' VGPRs: (\d+)'),('s',r'ScratchSize \[bytes/lane\]: (\d+)'),('sp',r'VGPRs Spill: (\d+)'),('l',r'LDS Size \[bytes/block\]: (\d+)'),('a',r'AGPRs: (\d+)')):
        m=re.search(p,l)
        if m and cur is not None: cur[k]=m.group(1)
for r in rows: print(f\"{r.get('v','?'):>4} v {r.get('a','?'):>4} a {r.get('s','?'):>5} scr {r.get('sp','?'):>5} spill {r.get('l','?'):>7} lds  {r['n']}\")
"
