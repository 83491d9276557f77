# Exploration session (GPU box): phase ablations and A/B builds of the one-wavefront-per-env kernels on configs 4 and 5.
O=gpurun_out/explore
mkdir -p $O
for w in "8192 synthetic50x20" "32768 mixed"; do
  for lib in shipped occ7 nodeepwalk; do
    if [ $lib = shipped ]; then unset JSSENV_AMD_LIB; else export JSSENV_AMD_LIB=$PWD/variants/$lib.so; fi
    JSS_NSUB=1,2 python tools/gpu_pipeline_probe.py $w 2>&1 | grep "n_sub\|lib="
  done
  export JSSENV_AMD_LIB=$PWD/variants/profiling.so
  for m in 0 1 2 4 8 16 31; do echo "== ablate mask $m"; JSS_NSUB=1,2 JSS_ABLATE=$m python tools/gpu_pipeline_probe.py $w 2>&1 | grep "n_sub"; done
  unset JSSENV_AMD_LIB
done > $O/wave_ablate.txt 2>&1
cat $O/wave_ablate.txt
