export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_teamdbg.so
python bench.py --steps 3 --warmup 1 --samples 10000 --batch-sites 100000 --no-cpu-baseline --no-rank-planes 2>&1 | grep "stream debug" | tail -7
