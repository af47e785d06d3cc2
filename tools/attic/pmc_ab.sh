#!/bin/bash
# PMC A/B of several builds (build/variants/*.so, made by tools/build_variant.sh) on the GPU box: VALU instructions and
# cycles per launch of the hot kernels under bench.py, one rocprofv3 --pmc pass per build (no other tracing domain).
# usage: tools/pmc_ab.sh <outdir>
out=${1:-gpurun_out/pmc_ab}
ROOT=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for lib in "$ROOT"/build/variants/*.so; do
  v=$(basename "$lib" .so)
  export D377_LIB=$lib
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$ROOT/$out/$v" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$ROOT/$out/$v.log" 2>&1
done
cd "$ROOT"
python3 - "$out" <<'P'
import csv, glob, collections, os, re, sys
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    v = os.path.basename(d.rstrip("/"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if not m: continue
            key = (m.group(1), r["Grid_Size"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
    for (k, g), c in sorted(agg.items()):
        if k.startswith("k_msm") or k.startswith("k_init"): continue
        n = len(cnt[(k, g)])
        print(v, k, "grid", g, "launches", n, "VALU/launch %.5g" % (c["SQ_INSTS_VALU"] / n), "cycles/XCD %.5g" % (c["GRBM_GUI_ACTIVE"] / n / 8))
P
