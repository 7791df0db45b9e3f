# the file leg twice in one process: staging page-locked afresh in the cold and in the warm run (same box)
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_ht WTAMD_BENCH_NO_HOSTDEC=1
python tools/e2e_bw_only.py 248.956422 > /dev/null 2>&1      # writes the files
for k in 1 2; do
  python tools/e2e_bw_only.py 248.956422 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cold %.3f s (submit %.0f ms, pinned %s device %s)  warm %.3f s (submit %.0f ms, pinned %s device %s)  steady %.3g' % (r['cold']['seconds'], r['cold']['host_submit_ms'], r['cold']['pinned_afresh'], r['cold']['device_afresh'], r['warm']['seconds'], r['warm']['host_submit_ms'], r['warm']['pinned_afresh'], r['warm']['device_afresh'], r['steady_bp_per_s']))"
done
