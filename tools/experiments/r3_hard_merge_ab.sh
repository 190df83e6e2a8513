for rep in 1 2; do
for lib in libbasevar_amd.so libbasevar_amd_prev.so; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  for cfg in "--samples 60 --batch-sites 1000000" "--samples 40 --batch-sites 1000000 --coverage 0.5" "--samples 1000 --batch-sites 500000 --coverage 0.05" "--samples 10000 --batch-sites 100000 --flags 16"; do
  python bench.py --steps 20 --warmup 3 $cfg --no-cpu-baseline | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('$lib $cfg: %.2f M sites/s  step %.4f ms  pass1 %.4f' % (d['value']/1e6, d['ms_per_step'], r['pass1_avg_ms']))"
  done
done
done
