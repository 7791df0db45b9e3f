# index kernel experiments: searched (default) vs scanned window index
for m in search scan; do
  echo "WTAMD_INDEX=$m"
  WTAMD_INDEX=$m python bench.py --no-cpu-baseline --no-e2e --no-sub --e2e-bw-mbp 0 --steps 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['index_kernel_ms'])"
done
