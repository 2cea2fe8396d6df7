# all configurations: profile passes + the bench line itself (GPU box).  ROUND=r04 (default) names the outputs: profiles/<ROUND>_<config>_*
set -e
ROUND=${ROUND:-r04}
mkdir -p gpurun_out/bench_$ROUND
for c in ${CONFIGS:-cornell mesh82k atrium mesh328k mixed spheres}; do
  case $c in cornell) A="";; *) A="--config $c --spp 8";; esac
  bash tools/profile.sh ${ROUND}_$c $A > gpurun_out/prof_$c.log 2>&1 || { tail -5 gpurun_out/prof_$c.log; }
  python tools/profile_summary.py gpurun_out/prof_${ROUND}_$c ${ROUND}_$c > gpurun_out/prof_${c}_summary.log 2>&1 || tail -3 gpurun_out/prof_${c}_summary.log
  cp profiles/${ROUND}_${c}_* gpurun_out/bench_$ROUND/ 2>/dev/null || true
  echo "profiled $c"
done
