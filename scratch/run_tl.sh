#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/tl; cd $R; QTOS_LIB=$L rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 scratch/one.py > /tmp/tl.log 2>&1
  echo "== $L"; python3 scratch/timeline.py /tmp/tl
done > $O/r3_tl.log 2>&1; cat $O/r3_tl.log
