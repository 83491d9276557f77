# Same-box A/B of library builds on tools/gpu_step_probe.py (jss_step with recorded actions, hipGraph replay).
# Usage (GPU box, through gpurun): bash tools/gpu_step_ab.sh <tag> "<lib1> <lib2> ..."   lib = "shipped" or a file under variants/
# (tools/build_instrumented.py: mstep6 / mstep7 = the fused grid's step kernel at 6 / 7 wavefronts per SIMD); gpurun_out/<tag>/step_probe.txt
cd $GRAFT_REPO_ROOT; TAG=${1:-r06w}; LIBS=${2:-"shipped mstep6 mstep7"}; mkdir -p gpurun_out/$TAG
for rep in 1 2; do for lib in $LIBS; do
  if [ $lib = shipped ]; then unset JSSENV_AMD_LIB; else export JSSENV_AMD_LIB=$PWD/variants/$lib.so; fi
  timeout 600 python tools/gpu_step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/$TAG/step_probe.txt
done; done
unset JSSENV_AMD_LIB
sort gpurun_out/$TAG/step_probe.txt
