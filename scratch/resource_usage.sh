#!/bin/bash
# usage: scratch/resource_usage.sh <file.hip> [regex]  -- VGPRs / occupancy / spills / LDS per kernel (gfx950)
f=$1; rx=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/ru/$(basename $f).o -Rpass-analysis=kernel-resource-usage $EXTRA 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = {'name': m.group(1)}; rows.append(cur); continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', l)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    if re.search(sys.argv[1], n):
        print('%-70s vgpr %3d agpr %3d occ %d spill %d lds %d scratch %d' % (n[:70], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('Occupancy', -1), r.get('VGPRs Spill', -1), r.get('LDS Size', -1), r.get('ScratchSize', -1)))
" "$rx"
