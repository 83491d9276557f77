# Round 5, GPU session D: the phase timeline of a wave's life (instrumented build) and the recorder's occupancy A/B
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
JSSENV_AMD_LIB=$R/variants/profiling.so timeout 600 python tools/gpu_wave_timeline.py > $O/wave_timeline.txt 2>&1; cat $O/wave_timeline.txt
for i in 1 2; do
for v in shipped traj2_5 traj2_4 allinplace; do
  L=$R/jssenv_amd/libjss_hip.so; [ $v != shipped ] && L=$R/variants/$v/libjss_hip.so
  echo "== $v" >> $O/traj_probe.txt
  JSSENV_AMD_LIB=$L timeout 300 python tools/gpu_traj_probe.py >> $O/traj_probe.txt 2>&1
done
done
grep -v amdgpu.ids $O/traj_probe.txt
