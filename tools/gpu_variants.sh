# time every build under variants/*.so with tools/gpu_ablate.py (first line = full kernel time)
for so in variants/*.so; do
  echo "== $so"
  for w in "65536 ta01" "16384 ta41" "16384 ta21"; do JSSENV_AMD_LIB=$PWD/$so python tools/gpu_ablate.py $w 2>&1 | grep -E "^full" | sed "s/^/$w /"; done
done
