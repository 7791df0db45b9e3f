# compares engine variants (csrc/build.py build_engine_variant) on the same box: C2 whole genome, kernel ms
# usage: tools/variants.sh <name> ...   ("-" = the default library)
for v in "$@"; do
  L=$PWD/wiggletools_amd/csrc/libwiggletools_amd.so
  [ "$v" != "-" ] && L=$PWD/wiggletools_amd/csrc/libwiggletools_amd_$v.so
  WTAMD_LIB=$L python bench.py --config ${CFG:-c2} --no-cpu-baseline --no-e2e --no-sub --e2e-bw-mbp 0 --steps 3 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); f=r['roofline']; print('$v', 'step_ms', round(r['ms_per_step'],2), 'kernel_ms', round(f['kernel_ms'],2), 'index_ms', round(f['index_kernel_ms'],2), 'frac', round(f['frac'],4))"
done
