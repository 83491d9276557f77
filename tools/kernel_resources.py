"""Registers, spills and scratch of every kernel in the built libjss_hip.so (from the code object notes).
usage: python tools/kernel_resources.py [lib.so] [substring filter]
`kernel_resources(so)` is what tests/test_abi_and_host.py reads: no kernel of the shipped library may use scratch memory."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(so):
    """[(demangled name, vgprs, sgprs, spilled vgprs, spilled sgprs, scratch bytes)] of the gfx950 code object inside `so`."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "jss.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = []
    for k in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", k).group(1))   # noqa: E731
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True, check=True).stdout.split("\n")
    clean = lambda n: n.replace("(anonymous namespace)::", "").replace("(jss::Params)", "").replace("void ", "")   # noqa: E731
    return sorted((clean(n),) + r[1:] for n, r in zip(names, rows))


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jssenv_amd", "libjss_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = kernel_resources(so)
    for n, v, s, vs, ss, scratch in rows:
        if flt in n:
            print(f"{n:44s} vgpr {v:3d} sgpr {s:3d} vspill {vs:3d} sspill {ss:3d} scratch {scratch}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r[5])} with scratch memory, {sum(1 for r in rows if r[3])} with VGPRs spilled")
