# the team form's ticket rules / start-up WITHOUT its helpers on the headline launch (library: make VARIANT=tk DEFS=-DBV_BIG_TICKET_RULES)
for rep in 1 2 3 4 5 6 7 8 9 10; do
for lib in libbasevar_amd.so libbasevar_amd_tk.so; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('$lib: %.3f M sites/s  step %.4f ms  pass1 %.4f ms = %.4f  pass2 %.4f' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['pass2_avg_launch_ms']))"
done
done
