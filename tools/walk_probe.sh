#!/bin/bash
# Median by walking vs the bitmap kernel on one real chromosome (chr21: index 20); optional: library variants given as arguments
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config c4 --chroms 20 --steps 3 --warmup 1 --no-e2e --no-sub --no-cpu-baseline --no-genome-files --e2e-bw-mbp 0 > gpurun_out/walk_$name.json 2> gpurun_out/walk_$name.err
  python - gpurun_out/walk_$name.json $name <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], "no line"); sys.exit()
r=json.loads(l[-1]); print(sys.argv[2], "ms/step", round(r.get('ms_per_step'),3), "frac", (r.get('roofline') or {}).get('frac'))
PY
  grep -h "wt_walk_profile" gpurun_out/walk_$name.err | tail -1
}
"$@"
