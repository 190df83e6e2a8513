# the engine's per-pass timing events on every launch against one launch in eight (bench.py --sparse-timing)
for rep in 1 2 3; do
for sp in "" "--sparse-timing"; do
  for cfg in "--samples 10000 --batch-sites 100000" "--samples 10000 --batch-sites 8192" "--batch-sites 8192" "--steps 24"; do
  python bench.py --steps 32 --warmup 5 $cfg $sp --no-cpu-baseline | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('[$sp] $cfg: %.2f M sites/s  step %.4f ms  %s avg %.4f ms = %.3f of peak (%d of %d launches timed)' % (d['value']/1e6, d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['launches_timed'], r['launches']))"
  done
done
done
