# all five configurations: profile passes + the bench line itself (GPU box)
set -e
mkdir -p gpurun_out/bench_r03
for c in ${CONFIGS:-cornell mesh82k mesh328k mixed spheres}; do
  case $c in cornell) A="";; mesh82k|mesh328k) A="--config $c --spp 8";; *) A="--config $c --spp 8";; esac
  bash tools/profile.sh r03_$c $A > gpurun_out/prof_$c.log 2>&1 || { tail -5 gpurun_out/prof_$c.log; }
  python tools/profile_summary.py gpurun_out/prof_r03_$c r03_$c > gpurun_out/prof_${c}_summary.log 2>&1 || tail -3 gpurun_out/prof_${c}_summary.log
  cp profiles/r03_${c}_* gpurun_out/bench_r03/ 2>/dev/null || true
  echo "profiled $c"
done
