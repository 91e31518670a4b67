#!/bin/bash
# round 5: the evidence of k_kkt5 on one box -- launch time with parts of a pair step compiled out (development libraries
# libqtos_abl<mask>.so: scratch/devbuild.sh abl<mask> -DQTOS_K5_ABL=<mask>), per-wave stamps of the diagnostic build
# (scratch/build.sh), and the same two for the walk's default kernel next to it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_experiments; mkdir -p $O; cd $R
{
  echo "# k_kkt5<128>, knots100 walk (100 stages = 50 pair steps), batch 256: ms per launch with parts of a pair step compiled out"
  echo "# (-DQTOS_K5_ABL bit mask: 1 assembly, 2 extraction, 4 Schur update, 8 next columns, 16 factor wave, 32 rhs chain, 64 V/W chain; wrong results)"
  python scratch/abl5.py abl0 abl1 abl2 abl4 abl8 abl16 abl32 abl64 abl127
  echo "# the same libraries' k_kkt2<112> (QTOS_KKT=2; no ablation switch reaches it: the box's reference)"
  QTOS_KKT=2 python scratch/abl5.py abl0
} > $O/kkt5_ablation.log 2>&1
QTOS_KKT=6 timeout 300 python scratch/stamps5.py > $O/kkt5_stamps.log 2>&1
QTOS_KKT=6 GAIT=trot timeout 300 python scratch/stamps5.py > $O/kkt5_stamps_trot.log 2>&1
cat $O/kkt5_ablation.log; tail -16 $O/kkt5_stamps.log
