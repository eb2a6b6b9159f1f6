"""Static opcode histogram of one kernel in a hipcc -save-temps .s file:  isa_hist.py file.s mangled-name-substring [top]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
c = collections.Counter()
for l in lines[start + 1:end]:
    m = re.match(r"\s+([a-z][a-z0-9_]+)\b", l)
    if m:
        c[m.group(1)] += 1
tot = sum(c.values())
groups = collections.Counter()
for k, v in c.items():
    g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else
         "vmem" if k.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other")
    groups[g] += v
print(lines[start].split(":")[0], "instructions:", tot, dict(groups))
for k, v in c.most_common(top):
    print("  %-28s %6d" % (k, v))
