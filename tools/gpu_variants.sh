# time every build under variants/*.so with tools/gpu_ablate.py (first line = full kernel time)
for so in variants/*.so; do
  echo "== $so"
  for w in "65536 ta01" "131072 ta01" "4096 ta01"; do JSS_KERNEL=$JSS_KERNEL JSSENV_AMD_LIB=$PWD/$so python tools/gpu_ablate.py $w 2>&1 | grep -E "^full" | sed "s/^/$w /"; done
done
