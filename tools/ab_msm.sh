#!/bin/bash
# Same-box A/B of library variants (build/variants/<name>.so) on the MSM of Elements: whole call at several sizes, alternating.
# usage: tools/ab_msm.sh <outfile> <variant> <variant> [...]
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2 3; do
  for v in "$@"; do
    echo "== $v rep $rep" >> "$out"
    D377_LIB=$PWD/build/variants/$v.so timeout -k 10 200 python3 tools/size_sweep.py --sizes 16384,65536,262144,1048576,4194304 --ops "msm (Elements)" 2>&1 | grep "n=" >> "$out"
  done
done
python3 - "$out" <<'P'
import re, sys, collections
d = collections.defaultdict(list); cur = None
for l in open(sys.argv[1]):
    m = re.match(r"== (\S+) rep", l)
    if m: cur = m.group(1); continue
    m = re.match(r"\s+n=2\^(\d+)\s+([\d.]+) us", l)
    if m: d[(int(m.group(1)), cur)].append(float(m.group(2)))
for (lg, v), xs in sorted(d.items()):
    print("n=2^%-2d %-10s %s  mean %.1f us" % (lg, v, " ".join("%8.1f" % x for x in xs), sum(xs) / len(xs)))
P
