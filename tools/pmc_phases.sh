#!/bin/bash
# Hardware counters of the penalty kernel (uph_eval_batch) attributed to the phases of an objective evaluation: the counter set of tools/pmc_eval.sh
# on the shipped library and on four diagnostic builds that leave one phase out (-DUPH_PHASE_MASK, solver_program.hpp: m14 without generate +
# expand, m13 without the samples, m11 without the scatter, m7 without the adjoint); phase = all - (all but it).
# usage (GPU box): bash tools/pmc_phases.sh <tag> [set]      builds: tools/build_variants.sh m14="-DUPH_PHASE_MASK=14" m13=... m11=... m7=...
TAG=$1; SET=${2:-lds}
cd $GRAFT_REPO_ROOT
for v in default m14 m13 m11 m7; do
  bash tools/pmc_eval.sh $TAG $v $SET > /dev/null 2>&1
done
python - $GRAFT_REPO_ROOT/gpurun_out/$TAG $SET <<'PY'
import sys, re, os
d, st = sys.argv[1], sys.argv[2]
def rd(v):
    out = {}
    p = os.path.join(d, "pmc_eval_%s_%s.txt" % (v, st))
    if not os.path.exists(p):
        return out
    for ln in open(p):
        m = re.match(r"(\S+)\s+(\S+)\s+\(dispatches (\d+)\)\s+per trajectory-evaluation (\S+)", ln)
        if m:
            out[m.group(1)] = float(m.group(4))
    return out
full = rd("default")
names = {"m14": "generate+expand", "m13": "samples", "m11": "scatter", "m7": "adjoint"}
print("per trajectory-evaluation (B = 8192 hill problems, 20 evaluations per launch); phase = shipped library - build without that phase")
print("%-24s %12s" % ("counter", "all phases") + "".join("%18s" % n for n in names.values()) + "%12s" % "rest")
for k in sorted(full):
    row, acc = "%-24s %12.1f" % (k, full[k]), 0.0
    for v in names:
        o = rd(v)
        ph = full[k] - o.get(k, float("nan"))
        acc += ph
        row += "%18.1f" % ph
    print(row + "%12.1f" % (full[k] - acc))
PY
