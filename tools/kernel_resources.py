"""Registers, spills and scratch of every kernel in the built libjss_hip.so (from the code object notes).
usage: python tools/kernel_resources.py [lib.so] [substring filter]"""
import subprocess,re,sys
so=sys.argv[1] if len(sys.argv)>1 else '/root/repo/jssenv_amd/libjss_hip.so'
subprocess.run(['objcopy','-O','binary','--only-section=.hip_fatbin',so,'/tmp/fat.bin'],check=True)
subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler','--unbundle','--type=o','--input=/tmp/fat.bin','--targets=hipv4-amdgcn-amd-amdhsa--gfx950','--output=/tmp/jss.co'],check=True)
r=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes','/tmp/jss.co'],capture_output=True,text=True).stdout
ks=r.split('- .agpr_count')
rows=[]
for k in ks[1:]:
    name=re.search(r'\.name:\s+(\S+)',k).group(1)
    g=lambda key: int(re.search(r'\.'+key+r':\s+(\d+)',k).group(1))
    rows.append((name,g('vgpr_count'),g('sgpr_count'),g('vgpr_spill_count'),g('sgpr_spill_count'),g('private_segment_fixed_size')))
names=subprocess.run(['c++filt']+[r[0] for r in rows],capture_output=True,text=True).stdout.split('\n')
flt=sys.argv[2] if len(sys.argv)>2 else ''
for n,r in sorted(zip(names,rows)):
    n=n.replace('(anonymous namespace)::','').replace('(jss::Params)','').replace('void ','')
    if flt in n: print(f"{n:44s} vgpr {r[1]:3d} sgpr {r[2]:3d} vspill {r[3]:3d} sspill {r[4]:3d} scratch {r[5]}")
