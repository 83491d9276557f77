"""Histogram of rocprofv3 PC samples per code-object offset (GPU box; tools/gpu_pcsample.sh).
usage: python tools/pcsample_hist.py pc_sampling.csv [kernel_trace.csv]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
cols = rows.fieldnames
print("columns:", cols)
off_col = next((c for c in cols if "offset" in c.lower()), None)
obj_col = next((c for c in cols if "code_object" in c.lower() and "id" in c.lower()), None)
inst_col = next((c for c in cols if c.lower() in ("instruction", "inst", "instruction_comment")), None)
extra = [c for c in cols if any(k in c.lower() for k in ("stall", "reason", "issued", "type", "wave_count", "snapshot"))]
hist, total = collections.Counter(), 0
reasons = collections.defaultdict(collections.Counter)
for r in rows:
    key = (r.get(obj_col), r.get(off_col))
    hist[key] += 1
    total += 1
    for c in extra:
        reasons[c][r[c]] += 1
print("samples:", total, " distinct pcs:", len(hist))
for c, cnt in reasons.items():
    print(c, cnt.most_common(12))
per_obj = collections.Counter()
for (o, _), n in hist.items():
    per_obj[o] += n
print("per code object:", per_obj.most_common(6))
print("top offsets (code object, offset, samples, %):")
for (o, off), n in hist.most_common(400):
    try:
        offs = hex(int(off))
    except Exception:
        offs = off
    print(o, offs, n, f"{100.0 * n / total:.2f}")
