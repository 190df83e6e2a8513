# the team form on the headline batch (libraries: make VARIANT=teamall DEFS=-DBV_TEAM_MAX_SITES=100000000; teamallp adds -DBV_TEAM_PRIO=2)
for rep in 1 2; do
for lib in libbasevar_amd.so libbasevar_amd_teamall.so libbasevar_amd_teamallp.so; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  for bs in 32768 131072; do
    echo "== $lib sites $bs"
    python bench.py --steps 20 --warmup 3 --samples 100000 --batch-sites $bs --no-cpu-baseline | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done
done
