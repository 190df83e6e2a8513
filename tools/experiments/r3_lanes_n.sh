cd "$(dirname "$0")/../.."
run() { timeout 300 python3 bench.py --no-cpu-baseline --warmup 4 --steps 24 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lanes env %s  %-50s sites/s %.4g ms/step %.4f' % ('$BASEVAR_AMD_LANES', '$*', d['value'], d['ms_per_step']))"; }
for n in 2 3 4; do
  export BASEVAR_AMD_LANES=$n
  run --samples 10000 --batch-sites 100000 --lanes 2 --distinct-batches 4
  run --samples 10000 --batch-sites 32768 --lanes 2 --distinct-batches 4
done
