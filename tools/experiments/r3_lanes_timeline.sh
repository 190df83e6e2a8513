# In-kernel timeline of the two lanes (no profiler in the way): library built with  make VARIANT=tl DEFS=-DBV_TL_DEBUG
# usage: r3_lanes_timeline.sh [runs]   -- prints, per run, sites/s and the last submits of both lanes on one time axis (us)
export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_tl.so
for rep in $(seq 1 ${1:-6}); do
  python bench.py --steps 30 --warmup 5 --samples 10000 --batch-sites 100000 --no-cpu-baseline --lanes 2 $TL_EXTRA > /tmp/tl_out.txt 2> /tmp/tl_err.txt
  python - <<'PY'
import json,re
d=json.loads([l for l in open('/tmp/tl_out.txt') if l.startswith('{')][0])
rows=[]
for l in open('/tmp/tl_err.txt'):
    m=re.match(r'\[timeline\] engine (\S+) submit (\d+): stream (\d+) (\d+)  solve16 (\d+) (\d+)  pass2 (\d+) (\d+)  wave-solver (\d+) (\d+)',l)
    if m: rows.append((m.group(1),int(m.group(2)))+tuple(int(x) for x in m.groups()[2:]))
eng=sorted(set(r[0] for r in rows))
rows=[r for r in rows if r[2] not in (0,0xFFFFFFFF)]
rows.sort(key=lambda r:r[2])
rows=rows[-12:]
t0=rows[0][2]
print("== %.1f M sites/s, step %.4f ms" % (d['value']/1e6, d['ms_per_step']))
for r in rows:
    f=lambda x:(x-t0)/100.0
    print("  lane %d submit %2d: stream %7.1f-%7.1f  solve16 %7.1f-%7.1f  wave-solver %7.1f-%7.1f  pass2 %7.1f-%7.1f" % (eng.index(r[0]), r[1], f(r[2]),f(r[3]),f(r[4]),f(r[5]),f(r[8]),f(r[9]),f(r[6]),f(r[7])))
PY
done
