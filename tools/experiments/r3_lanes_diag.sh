cd "$(dirname "$0")/../.."
run() { a=("$@"); [ "${a[-1]}" = "(null stream)" ] && unset "a[-1]"; timeout 300 python3 bench.py --no-cpu-baseline --warmup 3 "${a[@]}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-70s sites/s %.4g ms/step %.4f' % ('$*', d['value'], d['ms_per_step']))"; }
run --samples 10000 --batch-sites 8192 --lanes 2 --steps 60
BASEVAR_BENCH_NULLSTREAM=1 run --samples 10000 --batch-sites 8192 --lanes 2 --steps 60 "(null stream)"
run --samples 10000 --batch-sites 8192 --streams 2 --steps 60
run --samples 10000 --batch-sites 8192 --lanes 1 --steps 60
run --samples 10000 --batch-sites 32768 --lanes 2 --steps 60
run --samples 10000 --batch-sites 32768 --lanes 1 --steps 60
