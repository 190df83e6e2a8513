# the team form of the long-row kernel (default up to 32,768 sites) against the plain form (flags 256 = shape 1) on small batches;
# whole path (both passes) and pass 1 alone
for bs in 1024 2048 4096 8192 16384 32768; do
  for fl in 0 256; do
    python bench.py --steps 30 --warmup 5 --samples 100000 --batch-sites $bs --no-cpu-baseline --flags $fl | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('sites %6d  %s  %.2f M sites/s  step %.3f ms  pass 1 %.3f ms = %.3f of peak' % ($bs, 'plain' if $fl else 'team ', d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac']))"
  done
done
