#!/bin/bash
# Same-box A/B of fixed-base builds (build/variants/*.so): rate at 2^20 and 2^22, then TCC hit/miss and FETCH/WRITE per
# launch of k_scalar_mul_base (separate rocprofv3 --pmc passes, no other tracing domain).
# usage: tools/fb_ab.sh <outdir> variant...
out=$1; shift
ROOT=$(pwd)
mkdir -p "$out"
for v in "$@"; do
  echo "== $v"
  D377_LIB=$ROOT/build/variants/$v.so timeout 300 python3 tools/size_sweep.py --ops scalar_mul_base --sizes 1048576,4194304 2>&1 | grep "n="
done
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export D377_LIB=$ROOT/build/variants/$v.so
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$out/$v/pmc$i" -- python3 "$ROOT/tools/size_sweep.py" --ops scalar_mul_base --sizes 1048576 > "$ROOT/$out/$v.pmc$i.log" 2>&1
  done
done
cd "$ROOT"
python3 - "$out" "$@" <<'P'
import csv, glob, os, re, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    agg = collections.defaultdict(float); disp = set()
    for f in glob.glob(os.path.join(out, v, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_scalar_mul_base" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add((f, r["Dispatch_Id"]))
    per = collections.defaultdict(set)
    for f, d in disp: per[f].add(d)
    n = max(1, max(len(x) for x in per.values())) if per else 1
    c = {k: a / n for k, a in agg.items()}
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print("%s: k_scalar_mul_base 2^20, per launch (%d launches): FETCH %.1f MiB x2 (gfx950 correction) = %.0f B/element, WRITE %.1f MiB, "
          "TCC hit %.3g miss %.3g (hit rate %.3f), VALU %.4g, cycles/XCD %.4g, SQ_WAIT_ANY/SQ_WAVE_CYCLES %.3f" % (
          v, n, c.get("FETCH_SIZE", 0) / 1024, 2 * c.get("FETCH_SIZE", 0) * 1024 / (1 << 20), c.get("WRITE_SIZE", 0) / 1024, hit, miss,
          hit / max(1.0, hit + miss), c.get("SQ_INSTS_VALU", 0), c.get("GRBM_GUI_ACTIVE", 0) / 8,
          c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0))))
P
