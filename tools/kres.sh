#!/bin/bash
# usage: tools/kres.sh file.hip  -> one line per kernel: name vgprs scratch occupancy lds
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Rpass-analysis=kernel-resource-usage "$1" -o /tmp/kres.o 2>&1 | \
python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()[:110]}; rows.append(cur); continue
    for key,pat in (("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)"),("sgpr",r" SGPRs: (\d+)")):
        m=re.search(pat,line)
        if m and cur is not None: cur[key]=m.group(1)
    if "error" in line: print(line.rstrip())
for r in rows:
    print("%-112s v=%s a=%s s=%s scratch=%s occ=%s lds=%s"%(r["name"],r.get("vgpr"),r.get("agpr"),r.get("sgpr"),r.get("scratch"),r.get("occ"),r.get("lds")))
'
