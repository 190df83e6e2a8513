for rep in 1 2; do
for fl in 0 16384 20480; do
  for l in 1 2; do
  python bench.py --steps 30 --warmup 5 --samples 10000 --batch-sites 100000 --no-cpu-baseline --lanes $l --flags $fl | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('flags $fl lanes $l: %.2f M sites/s  step %.4f ms  stream %.4f' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms']))"
  done
done
done
