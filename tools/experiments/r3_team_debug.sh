# Life of the workgroups of a small long-row launch (team form of bv_pass1_kernel): when they start, stream their first row,
# run out of rows, finish solving -- distribution over workgroups and mean per XCD -- and how many solves were team jobs.
# Needs the instrumented library:  make -C basevar_amd/csrc VARIANT=teamdbg DEFS=-DBV_TEAM_DEBUG
export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_teamdbg.so
for bs in 1024 2048 8192; do
  echo "== sites $bs x 100000 samples"
  python bench.py --steps 2 --warmup 1 --samples 100000 --batch-sites $bs --no-cpu-baseline --no-rank-planes 2>&1 | grep "team debug" | tail -7
done
# short rows: when the waves of the streaming kernel finish their static ranges of sites
for bs in 100000; do
  echo "== sites $bs x 10000 samples"
  python bench.py --steps 3 --warmup 1 --samples 10000 --batch-sites $bs --no-cpu-baseline --no-rank-planes 2>&1 | grep "stream debug" | tail -7
done
