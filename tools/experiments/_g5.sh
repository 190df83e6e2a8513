cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do timeout 900 python -m pytest tests/test_bench_multirank.py -q -m gpu -k "tile_job_ranks_one_device" 2>&1 | tail -12 | cut -c1-700; done
timeout 2700 python -m pytest tests -x -q -m gpu --deselect tests/test_bench_multirank.py 2>&1 | tail -15 | cut -c1-500
