#!/bin/bash
# Same-box comparison of SEVERAL builds of the library (build/variants/*.so, the product): the same size sweep for each, the
# whole list forwards and then backwards (A B C ... C B A), means per build and size.
# usage (GPU box): tools/ab_many.sh out.txt "<sizes>" "<ops>" lib1.so lib2.so [lib3.so ...]
set -u
out=$1; SIZES=$2; OPS=$3; shift 3
libs=()
for l in "$@"; do
  [ -f "$l" ] || { echo "ab_many: $l is missing" >&2; exit 1; }
  libs+=("$(realpath "$l")")
done
: > "$out"
order=("${libs[@]}")
for ((i=${#libs[@]}-1; i>=0; i--)); do order+=("${libs[$i]}"); done
for t in "${order[@]}"; do
  echo "=== lib $t" >> "$out"
  D377_LIB=$t timeout -k 10 400 python3 -u tools/size_sweep.py --sizes $SIZES --ops "$OPS" > "$out.pass" 2>&1
  rc=$?
  grep -v "amdgpu.ids\|^one MI355X" "$out.pass" | sed 's/   graph:.*//' >> "$out"
  [ $rc -eq 0 ] && grep -q " us " "$out.pass" || { echo "ab_many: the sweep of $t failed (rc=$rc) or printed no timings" >&2; rm -f "$out.pass"; exit 1; }
  rm -f "$out.pass"
done
python3 - "$out" "${libs[@]}" <<'PY'
import re, sys, collections
t = open(sys.argv[1]).read()
libs = sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
tree = op = None
for l in t.splitlines():
    m = re.match(r"=== lib (\S+)", l)
    if m: tree = m.group(1); continue
    if l and not l.startswith(" "): op = l.strip(); continue
    m = re.match(r"\s+n=(\S+)\s+([0-9.]+) us", l)
    if m: res[(op, m.group(1))][tree].append(float(m.group(2)))
short = lambda p: p.split("/")[-1]
print("\nsummary: us per call, mean of the two passes (both passes), relative to %s" % short(libs[0]))
for (op, n), v in res.items():
    base = sum(v[libs[0]]) / len(v[libs[0]])
    print("  %-24s n=%-8s" % (op, n) + "".join("  %s %9.1f (%s) x%.3f" % (short(l), sum(v[l]) / len(v[l]), " ".join("%.1f" % x for x in v[l]), sum(v[l]) / len(v[l]) / base) for l in libs))
PY
