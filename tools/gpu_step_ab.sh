# Same-box A/B of library builds on tools/gpu_step_probe.py (jss_step with recorded actions, hipGraph replay): gpurun_out/r06w/step_probe.txt
# (libs: shipped + variants/mstep6.so, mstep7.so = the same sources with multi_min_blocks(kStep) in jss_kernels.hip set to 6 / 7)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06w
for rep in 1 2; do for lib in shipped mstep6 mstep7; do
  if [ $lib = shipped ]; then unset JSSENV_AMD_LIB; else export JSSENV_AMD_LIB=$PWD/variants/$lib.so; fi
  timeout 600 python tools/gpu_step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06w/step_probe.txt
done; done
sort gpurun_out/r06w/step_probe.txt
