for rep in 1 2 3; do
for lib in libbasevar_amd.so libbasevar_amd_team64.so; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  python bench.py --steps 20 --warmup 3 --samples 100000 --batch-sites 65536 --no-cpu-baseline | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('$lib', '%.2f M sites/s  step %.3f ms  pass 1 %.3f ms = %.3f of peak' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac']))"
done
done
