# After tools/gpu_profile_all.sh (through gpurun): copy the summaries of gpurun_out/prof_<tag>/ into profiles/<round>_<tag>/ and
# recompute profiles/hbm_traffic.json and the round's table in profiles/README.md.   usage: bash tools/copy_profiles.sh r06
cd "$(dirname "$0")/.."; R=${1:?round, e.g. r06}
for d in gpurun_out/prof_*/; do
  tag=$(basename $d); tag=${tag#prof_}; mkdir -p profiles/${R}_$tag
  cp $d/kernel_stats.csv $d/pmc_*_summary.csv profiles/${R}_$tag/
done
python tools/make_traffic_json.py $R && python tools/profile_table.py $R --write
