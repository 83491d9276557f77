cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_abl; mkdir -p $R/gpurun_out/pmc_abl
rocprofv3 -f csv --kernel-include-regex "jss_packed_kernel<16, 5>" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/pmc_abl -o abl -- python $R/tools/gpu_pmc_ablate.py > $R/gpurun_out/pmc_abl/log.txt 2>&1
python - <<'PY'
import csv, collections, os, statistics
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
rows=list(csv.DictReader(open(R+'/gpurun_out/pmc_abl/abl_counter_collection.csv')))
by=collections.defaultdict(dict)
for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(by)
names=['full','-check_no_op','-prioritize','-obs','-select','-advance','-all']
print('dispatches', len(ids))
base=None
for k,name in enumerate(names):
    chunk=ids[k*30:(k+1)*30]
    if not chunk: break
    med={c: statistics.median(by[i][c] for i in chunk) for c in by[chunk[0]]}
    if base is None: base=med
    print(f"{name:14s} VALU/wave {med['SQ_INSTS_VALU']/16384:7.1f} ({(med['SQ_INSTS_VALU']-base['SQ_INSTS_VALU'])/16384:+7.1f})  SALU/wave {med['SQ_INSTS_SALU']/16384:6.1f} ({(med['SQ_INSTS_SALU']-base['SQ_INSTS_SALU'])/16384:+6.1f})  LDS/wave {med['SQ_INSTS_LDS']/16384:5.1f}  active_valu {med['SQ_ACTIVE_INST_VALU']:.3g} wave_cycles {med['SQ_WAVE_CYCLES']:.3g}")
PY
find $R/gpurun_out/pmc_abl -name "*.csv" -size +1M -delete
