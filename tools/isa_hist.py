"""opcode histogram of an address range of one kernel (offsets relative to the kernel start, as tools/isa_blocks.py prints them)
usage: python tools/isa_hist.py all.s <kernel substring> 0xLO 0xHI"""
import collections, re, sys
path, key, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3], 16), int(sys.argv[4], 16)
lines = open(path).read().split("\n")
start = next(i for i, ln in enumerate(lines) if ln.endswith(">:") and key in ln)
base = None
h = collections.Counter()
for ln in lines[start + 1:]:
    if ln.endswith(">:"):
        break
    m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
    if not m:
        continue
    a = int(m.group(3), 16)
    if base is None:
        base = a
    if lo <= a - base < hi:
        h[m.group(1)] += 1
tot = sum(h.values())
print("total", tot)
for k, v in h.most_common(60):
    print("%-28s %5d" % (k, v))
