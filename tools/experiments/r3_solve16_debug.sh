# life of the workgroups of bv_p1s_solve16_kernel (library: make VARIANT=s16dbg DEFS="-DBV_TEAM_DEBUG -DBV_SOLVE16_DEBUG")
export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_s16dbg.so
python bench.py --steps 3 --warmup 1 --samples 10000 --batch-sites ${1:-100000} --no-cpu-baseline --no-rank-planes 2>&1 | grep "solve16 debug" | tail -6
