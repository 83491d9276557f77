# time every build under gpurun_variants/*.so with tools/gpu_ablate.py (first line = full kernel time)
for so in variants/*.so; do
  echo "== $so"
  JSSENV_AMD_LIB=$PWD/$so python tools/gpu_ablate.py 65536 ta01 2>&1 | grep -E "^full|eager"
  JSSENV_AMD_LIB=$PWD/$so python tools/gpu_ablate.py 16384 ta41 2>&1 | grep -E "^full"
done
