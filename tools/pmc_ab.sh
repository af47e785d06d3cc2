#!/bin/bash
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in a_old b_new; do
  export D377_LIB=$ROOT/build/variants/$v.so
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/s10/$v -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $ROOT/gpurun_out/s10/$v.log 2>&1
done
cd $ROOT
python3 - <<'P'
import csv, glob, collections
for v in ("a_old","b_new"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for f in glob.glob("gpurun_out/s10/%s/**/*counter_collection.csv"%v, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-40:]
            agg[(k,r["Grid_Size"])][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Grid_Size"])].add(r["Dispatch_Id"])
    for (k,g),c in sorted(agg.items()):
        if "k_" in k and ("roundtrip" in k or "sqrt" in k or "encode" in k or "scalar_mul_var" in k or "decompress" in k):
            n=len(cnt[(k,g)])
            print(v, k, g, "launches",n, "VALU/launch %.4g"%(c["SQ_INSTS_VALU"]/n), "cycles/XCD %.4g"%(c["GRBM_GUI_ACTIVE"]/n/8))
P
