# quick look at the register-column reducers (median, wilcoxon) on chromosome 21 (46.7 Mbp): step ms, per 31 Mbp
for c in c4 c5; do
  WTAMD_LIB=${WTAMD_LIB:-$PWD/wiggletools_amd/csrc/libwiggletools_amd.so} python bench.py --config $c --chroms 20 --no-cpu-baseline --no-e2e --no-sub --e2e-bw-mbp 0 --steps 2 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); f=r['roofline']; print('$c', 'step_ms', round(r['ms_per_step'],2), 'kernel_ms', round(f['kernel_ms'],2), 'per31Mbp', round(r['ms_per_step']*31/46.709983,2), 'frac', round(f['frac'],4), 'auc', r.get('auc_check'))"
done
