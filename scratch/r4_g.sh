#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-n}; shift
AB_KKT=${AB_KKT:-4} AB_GAITS=walk,trot timeout 1500 python scratch/ab4.py "$@" 2>&1 | grep -v "^qtos\|amdgpu" > $O/r4_ab_$T.log
python - <<PY
import re, collections
d = collections.defaultdict(list)
for l in open("$O/r4_ab_$T.log"):
    m = re.match(r'(\S+) .*?KKT=(\d) (\w+) kkt ms/launch ([0-9.]+)', l)
    if m: d[(m.group(1), m.group(3), m.group(2))].append(float(m.group(4)))
for k in sorted(d): print(k, d[k])
PY
