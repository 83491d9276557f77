"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -save-temps).
usage: python tools/isa_mix.py file.s [substring-of-mangled-name ...]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith(("s_load", "s_buffer")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    pats = sys.argv[2:]
    cur, mix, ops = None, {}, {}
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            mix[cur], ops[cur] = collections.Counter(), collections.Counter()
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        t = ln.strip()
        if not t or t[0] in ".;/" or t.endswith(":"):
            continue
        op = t.split()[0]
        mix[cur][classify(op)] += 1
        ops[cur][op] += 1
    for name in mix:
        if pats and not any(p in name for p in pats):
            continue
        print(name, sum(mix[name].values()), dict(mix[name]))
        if pats:
            print("   ", ops[name].most_common(30))


if __name__ == "__main__":
    main()
