cd $GRAFT_REPO_ROOT
for v in g32t1 g16t2 g16t4 g64t2 g64t1; do
  echo "== $v"
  bash tools/budget_sweep.sh libvr_hip_$v.so 128
  CONFIG=c1 bash tools/budget_sweep.sh libvr_hip_$v.so 8 32
  CONFIG=c2 bash tools/budget_sweep.sh libvr_hip_$v.so 64
done
