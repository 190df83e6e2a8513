cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --samples 10000 --batch-sites 100000 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('n10k', round(d['value']/1e6,1), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['frac_covers'])"; done
timeout 300 python bench.py --no-cpu-baseline --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('n100k', round(d['value']/1e6,2), d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['pass2_avg_launch_ms'],3))"
timeout 300 python bench.py --no-cpu-baseline --samples 10000 --batch-sites 8192 --steps 40 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('n10k_8192', round(d['value']/1e6,1), d['roofline']['kernel'], round(d['roofline']['frac'],3))"
