#!/bin/bash
# usage: tools/perf.sh "<op> <tracks> <mean_run> [lib-variant]" ...   (runs on the GPU box)
for cfg in "$@"; do
  set -- $cfg
  LIBV=""
  if [ -n "$4" ]; then LIBV="wiggletools_amd/csrc/libwiggletools_amd_$4.so"; fi
  WTAMD_LIB=$LIBV timeout 300 python bench.py --scale ${SCALE:-0.01} --steps 3 --warmup 1 --op $1 --tracks $2 --mean-run $3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$cfg', 'bp/s %.3e'%d['value'], 'kernel_ms %.2f index_ms %.2f'%(r['kernel_ms'], r['index_kernel_ms']), 'runs', d['config']['output_runs_per_gpu'], 'W', d['config']['window_bp'], 'lds', d['config']['lds_bytes_per_workgroup'], 'algGB/s %.0f'%r['achieved'])
    elif 'rror' in l: print(l.strip())
"
done
