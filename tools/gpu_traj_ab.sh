# Same-box A/B of library builds on the kernels that loop over steps (tools/gpu_traj_probe.py): gpurun_out/<tag>/traj_probe.txt
# usage: bash tools/gpu_traj_ab.sh <tag> "<lib1> <lib2> ..."   (lib = shipped | a file under variants/)
cd ${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-traj_ab}; LIBS=${2:-shipped}; mkdir -p gpurun_out/$TAG
for rep in 1 2; do for lib in $LIBS; do
  if [ $lib = shipped ]; then unset JSSENV_AMD_LIB; else export JSSENV_AMD_LIB=$PWD/variants/$lib.so; fi
  timeout 600 python tools/gpu_traj_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/$TAG/traj_probe.txt
done; done
sort gpurun_out/$TAG/traj_probe.txt
