# interleaved A/B of this build against basevar_amd/lib/libbasevar_amd_prev.so (an older commit built aside) over four configurations
for rep in 1 2 3; do
for lib in libbasevar_amd.so libbasevar_amd_prev.so; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  for cfg in "--samples 10000 --batch-sites 100000" "--samples 10000 --batch-sites 8192" "--batch-sites 8192" "--steps 20"; do
  python bench.py --steps 30 --warmup 5 $cfg --no-cpu-baseline | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('$lib $cfg: %.2f M sites/s  step %.4f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
done
