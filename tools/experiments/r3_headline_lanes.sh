for rep in 1 2 3; do
for l in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --lanes $l | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('headline lanes $l: %.3f M sites/s  step %.4f ms  pass1 %.3f pass2 %.3f' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['pass2_avg_launch_ms']))"
done
done
