# configs[1], one and two lanes: this build against older libraries (basevar_amd/lib/libbasevar_amd_<name>.so), interleaved
for rep in 1 2 3 4; do
for lib in libbasevar_amd.so libbasevar_amd_old.so libbasevar_amd_static.so; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  for l in 1 2; do
  python bench.py --steps 30 --warmup 5 --samples 10000 --batch-sites 100000 --no-cpu-baseline --lanes $l | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('$lib lanes $l: %.2f M sites/s  step %.4f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
done
