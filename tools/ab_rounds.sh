#!/bin/bash
# Same-box A/B of two TREES of this repository (e.g. last round's final commit against HEAD): the same size sweep run in
# either tree, twice each, in the order A B B A (with the same copy of tools/size_sweep.py in both trees).  usage (on the GPU box, from the repo root):
#   tools/ab_rounds.sh build/r03tree . gpurun_out/ab_rounds.txt
# where build/r03tree holds `git archive <commit> | tar -x` with `make lib` run in it.
A=$1; B=$2; out=$3
SIZES=256,4096,65536,262144,1048576,4194304
OPS="msm (Elements),msm (Encodings),scalar_mul_var,scalar_mul_var_element,scalar_mul_base,sqrt_ratio_zeta,decompress,compress,encode_to_curve,hash_to_curve"
ROOT=$(pwd)
: > "$out"
for rep in 1 2; do
  if [ $rep = 1 ]; then order=("$A" "$B"); else order=("$B" "$A"); fi     # A B B A: neither tree always runs second
  for t in "${order[@]}"; do
    echo "=== tree $t, pass $rep" >> "$out"
    (cd "$t" && timeout -k 10 400 python3 tools/size_sweep.py --sizes $SIZES --ops "$OPS" 2>&1 | grep -v "amdgpu.ids\|^one MI355X" | sed 's/   graph:.*//') >> "$ROOT/$out"
  done
done
python3 - "$out" "$A" "$B" <<'P'
import re, sys, collections
t = open(sys.argv[1]).read()
res = collections.defaultdict(lambda: collections.defaultdict(list))
tree = op = None
for l in t.splitlines():
    m = re.match(r"=== tree (\S+),", l)
    if m: tree = m.group(1); continue
    if l and not l.startswith(" "): op = l.strip(); continue
    m = re.match(r"\s+n=(\S+)\s+([0-9.]+) us", l)
    if m: res[(op, m.group(1))][tree].append(float(m.group(2)))
A, B = sys.argv[2], sys.argv[3]
print("\nsummary: us per call, mean of the two passes, %s -> %s (ratio)" % (A, B))
for (op, n), v in res.items():
    if A in v and B in v:
        a, b = sum(v[A]) / len(v[A]), sum(v[B]) / len(v[B])
        print("  %-24s n=%-8s %9.1f -> %9.1f   x%.2f" % (op, n, a, b, b / a))
P
