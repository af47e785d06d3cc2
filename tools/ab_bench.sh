#!/bin/bash
# A/B several builds of libdecaf377_amd.so (build/variants/*.so) with bench.py on the GPU box.
# usage: tools/ab_bench.sh [bench args]
for lib in build/variants/*.so; do
  echo "== $lib"
  D377_LIB=$PWD/$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('var-base %.3e/s  ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v['per_sec'] / 1e6, 1) for k, v in d.get('extra', {}).items() if isinstance(v, dict)})
    else:
        print(l)
"
done
