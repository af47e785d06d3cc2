#!/bin/bash
# hipcc's per-kernel resource table (VGPRs, spills, scratch, LDS, occupancy) for both translation units.
# usage: tools/resource_usage.sh [extra -D flags]  > profiles/rNN_resource_usage.txt
for u in d377 msm codec_chunked batch_msm; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c decaf377_amd/csrc/$u.hip -o /tmp/ru_$u.o 2> /tmp/ru_$u.txt
  python3 - /tmp/ru_$u.txt <<'P'
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    m = re.search(r"k_[a-z0-9_]+", b.split()[0])
    print("%-26s SGPRs %3s  VGPRs %3s  AGPRs %s  scratch B/lane %3s  occupancy %s  SGPR spills %2s  VGPR spills %2s  LDS B/block %s" % (
        m.group(0) if m else b.split()[0][:26], g("SGPRs"), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]")))
P
done
