# two-lane bench processes alternating with processes of other configurations (the slow two-lane runs of round 3 all came right
# after a process of another configuration): how often does a two-lane process run slow?
for rep in $(seq 1 ${1:-12}); do
  python bench.py --steps 30 --warmup 5 --samples 10000 --batch-sites 100000 --no-cpu-baseline $( [ $((rep % 3)) = 0 ] && echo "--groups 2" ) > /dev/null 2>&1
  python bench.py --steps 30 --warmup 5 --samples 10000 --batch-sites 100000 --no-cpu-baseline --lanes 2 | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('lanes 2 (run $rep): %.1f M sites/s' % (d['value']/1e6))"
done
