# e2e legs against the number of host threads the library uses (cgroup quota: 16 cores on the GPU box)
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_ht WTAMD_BENCH_NO_HOSTDEC=1
python tools/e2e_bw_only.py 248.956422 > /dev/null 2>&1      # writes the files
for n in 16 12 10 8; do
  echo "WTAMD_HOST_THREADS=$n"
  WTAMD_HOST_THREADS=$n python tools/e2e_bw_only.py 248.956422 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  bigwig cold %.3g warm %.3g steady %.3g  open %.3f/%.3f' % (r['bp_per_s'], r['warm_bp_per_s'], r['steady_bp_per_s'], r['cold']['open_seconds'], r['warm']['open_seconds']))"
  WTAMD_HOST_THREADS=$n python tools/e2e_only.py 248.956 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  bulk %.3g steady %.3g  pop %.3g' % (r['bulk']['bp_per_s'], r['bulk']['steady_bp_per_s'], r['pop']['bp_per_s']))"
done
