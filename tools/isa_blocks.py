"""Static view of one kernel of libunevenhip.so: basic blocks with instruction-class counts, barriers / clock reads as phase markers, and
loops (backward branches).  usage: python tools/isa_blocks.py <all.s from llvm-objdump -d> <kernel name substring> [--min N]"""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "lane"
        return "valu"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith("s_memtime") or op.startswith("s_memrealtime"):
        return "clock"
    return "salu"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if ln.endswith(">:") and key in ln:
            start = i
            break
    assert start is not None, "kernel not found"
    ins = []
    base = None
    for ln in lines[start + 1:]:
        if ln.endswith(">:"):
            break
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if not m:
            continue
        addr = int(m.group(3), 16)
        if base is None:
            base = addr
        tgt = None
        mt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", ln)
        if mt and m.group(1).startswith(("s_cbranch", "s_branch")):
            tgt = int(mt.group(1), 16)
        ins.append((addr - base, m.group(1), m.group(2), tgt))
    leaders = {0}
    for k, (a, op, args, tgt) in enumerate(ins):
        if tgt is not None:
            leaders.add(tgt)
            if k + 1 < len(ins):
                leaders.add(ins[k + 1][0])
    blocks = []
    cur = None
    for a, op, args, tgt in ins:
        if a in leaders or cur is None:
            cur = dict(start=a, n=0, cnt={}, tgt=None, ops=[])
            blocks.append(cur)
        c = classify(op)
        cur["cnt"][c] = cur["cnt"].get(c, 0) + 1
        cur["n"] += 1
        cur["ops"].append((op, args))
        if tgt is not None:
            cur["tgt"] = tgt
            cur["br"] = op
    print("kernel: %d instructions, %d blocks" % (len(ins), len(blocks)))
    tot = {}
    for b in blocks:
        for k, v in b["cnt"].items():
            tot[k] = tot.get(k, 0) + v
    print("static totals:", tot)
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    for b in blocks:
        c = b["cnt"]
        mark = ""
        if c.get("barrier"):
            mark += " BARRIER x%d" % c["barrier"]
        if c.get("clock"):
            mark += " CLOCK x%d" % c["clock"]
        back = b["tgt"] is not None and b["tgt"] <= b["start"]
        if back:
            mark += " <-- LOOP back to 0x%x (%s)" % (b["tgt"], b["br"])
        elif b["tgt"] is not None:
            mark += " -> 0x%x (%s)" % (b["tgt"], b["br"])
        f64 = sum(1 for op, _ in b["ops"] if "f64" in op)
        trans = sum(1 for op, _ in b["ops"] if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div_", "v_trig", "v_frexp", "v_ldexp")))
        mov = sum(1 for op, _ in b["ops"] if op.startswith(("v_mov", "v_accvgpr", "v_cndmask")))
        if b["n"] >= minn or mark:
            print("0x%06x n=%4d valu=%4d (f64 %4d, trans %3d, mov/sel %3d) lane=%3d salu=%3d lds=%3d vmem=%3d smem=%2d wait=%3d%s" % (
                b["start"], b["n"], c.get("valu", 0), f64, trans, mov, c.get("lane", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0), c.get("smem", 0), c.get("wait", 0), mark))


if __name__ == "__main__":
    main()
