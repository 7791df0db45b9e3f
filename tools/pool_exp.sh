# warm-run cost of the file leg against the size of the pinned pool (same box)
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_ht WTAMD_BENCH_NO_HOSTDEC=1
python tools/e2e_bw_only.py 248.956422 > /dev/null 2>&1      # writes the files
for mb in 4096 16384 4096 16384; do
  WTAMD_PINNED_POOL_MB=$mb python tools/e2e_bw_only.py 248.956422 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pool $mb MB: cold %.3f s (submit %.0f ms)  warm %.3f s (submit %.0f ms)  steady %.3g' % (r['cold']['seconds'], r['cold']['host_submit_ms'], r['warm']['seconds'], r['warm']['host_submit_ms'], r['steady_bp_per_s']))"
done
