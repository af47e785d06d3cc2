#!/bin/bash
# Same-box A/B of two builds of the library (build/variants/*.so or the product), the same size sweep in the order A B B A.
# usage (GPU box): tools/ab_variants.sh build/variants/noprio.so decaf377_amd/lib/libdecaf377_amd.so out.txt [sizes] [ops]
set -u
A=$(realpath "$1"); B=$(realpath "$2"); out=$3
[ -f "$A" ] && [ -f "$B" ] || { echo "ab_variants: a library is missing ($A, $B)" >&2; exit 1; }
SIZES=${4:-524288,1048576,1572864,2097152,4194304}
OPS=${5:-"scalar_mul_var,scalar_mul_var_element,scalar_mul_base,sqrt_ratio_zeta,decompress,compress,roundtrip,encode_to_curve,hash_to_curve"}
: > "$out"
for rep in 1 2; do
  if [ $rep = 1 ]; then order=("$A" "$B"); else order=("$B" "$A"); fi
  for t in "${order[@]}"; do
    echo "=== lib $t, pass $rep" >> "$out"
    D377_LIB=$t timeout -k 10 400 python3 -u tools/size_sweep.py --sizes $SIZES --ops "$OPS" > "$out.pass" 2>&1
    rc=$?
    grep -v "amdgpu.ids\|^one MI355X" "$out.pass" | sed 's/   graph:.*//' >> "$out"
    [ $rc -eq 0 ] && grep -q " us " "$out.pass" || { echo "ab_variants: the sweep of $t failed (rc=$rc) or printed no timings" >&2; rm -f "$out.pass"; exit 1; }
    rm -f "$out.pass"
  done
done
python3 - "$out" "$A" "$B" <<'P'
import re, sys, collections
t = open(sys.argv[1]).read()
res = collections.defaultdict(lambda: collections.defaultdict(list))
tree = op = None
for l in t.splitlines():
    m = re.match(r"=== lib (\S+),", l)
    if m: tree = m.group(1); continue
    if l and not l.startswith(" "): op = l.strip(); continue
    m = re.match(r"\s+n=(\S+)\s+([0-9.]+) us", l)
    if m: res[(op, m.group(1))][tree].append(float(m.group(2)))
A, B = sys.argv[2], sys.argv[3]
print("\nsummary: us per call, the two passes of each, %s -> %s (ratio of means)" % (A, B))
for (op, n), v in res.items():
    if A in v and B in v:
        a, b = sum(v[A]) / len(v[A]), sum(v[B]) / len(v[B])
        print("  %-24s n=%-8s %9.1f %9.1f -> %9.1f %9.1f   x%.3f" % (op, n, v[A][0], v[A][-1], v[B][0], v[B][-1], b / a))
P
