#!/usr/bin/env python3
"""Static instruction mix of one kernel of a device assembly file (hipcc --cuda-device-only -S), split at s_barrier.
usage: isa_stats.py cagpu.s <substring of the mangled kernel name>"""
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
seg, segs = {}, []
tot = {}
def cls(op):
    if op.startswith("v_"):
        if "f64" in op: return "valu64"
        return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem"
    return "other"
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    c = cls(op)
    seg[c] = seg.get(c, 0) + 1
    tot[c] = tot.get(c, 0) + 1
    if op == "s_barrier":
        segs.append(seg); seg = {}
segs.append(seg)
keys = ["valu", "valu64", "salu", "lds", "vmem", "wait", "branch"]
print("seg  " + " ".join("%7s" % k for k in keys))
for i, s in enumerate(segs):
    print("%3d  " % i + " ".join("%7d" % s.get(k, 0) for k in keys))
print("tot  " + " ".join("%7d" % tot.get(k, 0) for k in keys))
m = re.search(r"; NumVgprs: (\d+)", "\n".join(lines[end:end + 400]))
for l in lines[end:end + 60]:
    if any(k in l for k in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize", "codeLenInByte")):
        print(l.strip())
