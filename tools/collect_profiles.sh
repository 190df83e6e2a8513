#!/bin/bash
# Runs on the MI355X box (through gpurun): everything the committed profiles/ summaries are made of.
#   tools/collect_profiles.sh <tag>
# then, back in the container:  python tools/summarize_profiles.py <tag> 131072 100000
TAG=${1:-r1}
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
O=$ROOT/gpurun_out
rm -rf $O/prof_${TAG}_stats $O/prof_${TAG}_fetch $O/prof_${TAG}_write
timeout 600 python3 bench.py > $O/bench_${TAG}.json 2> $O/bench_${TAG}.err
timeout 300 python3 bench.py --streams 2 --no-cpu-baseline > $O/bench_${TAG}_2streams.json 2>> $O/bench_${TAG}.err
timeout 300 python3 bench.py --samples 10000 --batch-sites 524288 --no-cpu-baseline > $O/bench_${TAG}_N10000.json 2>> $O/bench_${TAG}.err
timeout 300 python3 bench.py --samples 1000000 --batch-sites 16384 --steps 8 --no-cpu-baseline > $O/bench_${TAG}_N1000000.json 2>> $O/bench_${TAG}.err
timeout 300 python3 bench.py --tally-only --no-cpu-baseline > $O/bench_${TAG}_tallyonly.json 2>> $O/bench_${TAG}.err
timeout 300 python3 bench.py --batch-sites 32768 --no-cpu-baseline > $O/bench_${TAG}_32k_batches.json 2>> $O/bench_${TAG}.err
timeout 600 python3 bench.py --no-cpu-baseline --with-tile-mode --with-host-path --batch-sites 65536 --tile-sites 65536 --steps 3 --warmup 1 > $O/bench_${TAG}_tilemode_100k.json 2>> $O/bench_${TAG}.err
timeout 900 python3 bench.py --no-cpu-baseline --with-tile-mode --samples 1000000 --batch-sites 8192 --tile-sites 8192 --steps 3 --warmup 1 > $O/bench_${TAG}_tilemode_1M.json 2>> $O/bench_${TAG}.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -- python3 bench.py --no-cpu-baseline > $O/bench_${TAG}_profiled.json 2>> $O/bench_${TAG}.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/bench_${TAG}.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/bench_${TAG}.err
tail -c 600 $O/bench_${TAG}.json
