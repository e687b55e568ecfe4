# round 5 A/B: HEAD's two-wave kernel against the message-passing one; then per-wave timelines
# usage: bash tools/gpu/r05a.sh "libA.so libB.so" "tlA.so tlB.so"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/gpu/c3dev.sh $1 2>&1 | tee gpurun_out/r05/ab_a.txt
for lib in $2; do
  [ -f fixed-wing-gym_amd/gym_fixed_wing/_abl/$lib ] || continue
  echo "== $lib"
  TL_LIB=$lib TL_WL=c3:log TL_STAGGER=2000 TL_PERM=1 timeout 300 python tools/timeline.py run > gpurun_out/r05/timeline_$lib.txt 2>&1
  grep -v "draw stage\|per XCD\|last block\|^{\|^/\|return fnb\|episode end\|early lanes\|draw piece\|block [0-9]" gpurun_out/r05/timeline_$lib.txt | head -12
done
