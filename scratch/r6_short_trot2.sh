#!/bin/bash
# round 6: the three parts of order rule 1 one by one (experiment library libqtos_expo.so, -DQTOS_EXP_ORDER): accuracy of one KKT solve per configuration
R=$GRAFT_REPO_ROOT; cd $R
code=$(python3 - <<'PY'
import re
s = open("scratch/r6_short_trot.py").read()
print(re.search(r"code = '''(.*?)'''", s, re.S).group(1))
PY
)
for c in trot,2.5,0.05 trot,5.0,0.1 trot,2.5,0.1 trot,4.0,0.1 trot,5.0,0.05 trot,8.0,0.05 trot,10.0,0.1 walk,5.0,0.1 walk,2.5,0.05 walk,5.0,0.05 walk,8.0,0.1; do
  for b in $@; do
    echo -n "bits $b "; CFG=$c QTOS_LIB=libqtos_expo.so QTOS_ORDER=0 QTOS_EXP_ORDERBITS=$b python3 -c "$code" 2>&1 | grep -v amdgpu.ids
  done
done
