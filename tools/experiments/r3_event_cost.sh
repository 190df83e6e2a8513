# what the four timing events of a submit cost (BASEVAR_AMD_EXP_NOEVENTS=1 skips them; timing fields are then meaningless)
for rep in 1 2 3; do
for ne in 0 1; do
  if [ $ne = 1 ]; then export BASEVAR_AMD_EXP_NOEVENTS=1; else unset BASEVAR_AMD_EXP_NOEVENTS; fi
  for cfg in "--samples 10000 --batch-sites 100000" "--samples 10000 --batch-sites 8192" "--steps 20"; do
  python bench.py --steps 30 --warmup 5 $cfg --no-cpu-baseline | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('noevents=$ne $cfg: %.2f M sites/s  step %.4f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
done
