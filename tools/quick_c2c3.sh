# quick look at the two difference-array lines of the bench (C2 mean, C3 stddev): step, kernel and index ms
for c in c2 c3; do
  python bench.py --config $c --no-cpu-baseline --no-e2e --no-sub --e2e-bw-mbp 0 --steps 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); f=r['roofline']; print('$c', 'step_ms', round(r['ms_per_step'],2), 'kernel_ms', round(f['kernel_ms'],2), 'index_ms', round(f['index_kernel_ms'],2), 'frac', round(f['frac'],4))"
done
