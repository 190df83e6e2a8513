set -e
for fl in 0 2 4 6; do
  echo "== flags $fl"
  python bench.py --steps 30 --warmup 5 --samples 100000 --batch-sites 8192 --no-cpu-baseline --no-rank-planes --flags $fl | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
echo "== tally only"
python bench.py --steps 30 --warmup 5 --samples 100000 --batch-sites 8192 --no-cpu-baseline --no-rank-planes --tally-only | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
for bs in 2048 4096 16384; do
  echo "== sites $bs"
  python bench.py --steps 30 --warmup 5 --samples 100000 --batch-sites $bs --no-cpu-baseline --no-rank-planes | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
