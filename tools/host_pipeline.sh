#!/bin/bash
# Host pipeline end to end (bv_call): stage seconds and sites/s, the way the reference prints its phases.
#   tools/host_pipeline.sh <tag>        -> gpurun_out/host_pipeline_<tag>.txt  (+ .json lines)
# (a) the reference's 100-BAM test set (copied to tests/golden/_local/bam100 by `tools/host_pipeline.sh stage` in the build
#     container; all-N surrogate FASTA as in tools/real_data_campaign.py), BAM -> pileup -> engine -> VCF/CVG;
# (b) synthetic batchfiles, 10,000 samples in files of 200 (the reference's --batch-count): the producer alone on 1 .. 96 host
#     threads (parallel efficiency), then bv_call end to end on 1 .. 64.
cd "$(dirname "$0")/.."; ROOT=$PWD
if [ "$1" = "stage" ]; then
  mkdir -p tests/golden/_local/bam100
  cp /root/reference/tests/data/140k_thalassemia_brca_bam/bam100/*.bam* tests/golden/_local/bam100/ 2>/dev/null
  cp /root/reference/tests/data/140k_thalassemia_brca_bam/sample_group.info tests/golden/_local/bam100/
  ls tests/golden/_local/bam100 | wc -l
  exit 0
fi
TAG=${1:-r5}; SITES=${2:-12000}; shift; shift
mkdir -p gpurun_out; OUT=gpurun_out/host_pipeline_$TAG.txt; : > $OUT
W=$(mktemp -d)
CALL=basevar_amd/lib/bv_call
say() { echo "$*" | tee -a $OUT; }
# ---- (a) 100 BAMs
if ls tests/golden/_local/bam100/*.bam > /dev/null 2>&1; then
  python3 - "$W/nn.fa" <<'PY'
import sys
with open(sys.argv[1], "w") as f:
    for name, L in (("chr11", 5250000), ("chr17", 41280000)):
        f.write(">%s\n" % name); f.write(("N" * 60 + "\n") * (L // 60 + 1))
PY
  ARGS=""; for b in tests/golden/_local/bam100/*.bam; do ARGS="$ARGS -I $b"; done
  for t in "1 0" "1 100" "8 0" "8 100"; do
    set -- $t; t=$1; export BASEVAR_AMD_BAM_KEEP=$2
    say "== (a) 100 BAMs, chr11:5246595-5248428,chr17:41197764-41276135, --thread $t, BAM readers kept open: $BASEVAR_AMD_BAM_KEEP"
    $CALL $ARGS -R $W/nn.fa --regions chr11:5246595-5248428,chr17:41197764-41276135 --mapq 10 --min-af 0.05 --thread $t \
      --pop-group tests/golden/_local/bam100/sample_group.info --output-vcf $W/a.vcf --output-cvg $W/a.cvg --timing $W/a.json 2>&1 | tail -2 | tee -a $OUT
    cat $W/a.json >> $OUT
  done
else
  say "(a) skipped: tests/golden/_local/bam100 not staged"
fi
# ---- (b) synthetic batchfiles
g++ -O2 -std=c++17 tools/gen_batchfiles.cpp -lz -o $W/gen || exit 1
g++ -O2 -std=c++17 -pthread -I include tools/producer_bench.cpp -lz -o $W/pbench || exit 1
mkdir -p $W/bf; $W/gen $W/bf 10000 200 $SITES 0.08 7
BF=$(ls $W/bf/*.gz | paste -sd,)
QUOTA=$(python3 -c "
try:
    q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
    print(0 if q == 'max' else float(q) / float(p))
except Exception:
    print(0)")
say "== (b) synthetic batchfiles: 10000 samples in $(ls $W/bf | wc -l) files, $SITES sites, $(du -sh $W/bf | cut -f1) BGZF; host: $(nproc) logical CPUs, container CPU quota (cgroup cpu.max): $QUOTA CPUs (0 = none)"
say "   (parallel efficiency is quoted against min(threads, quota): threads beyond the quota only take turns on the same CPU time)"
say "-- the producer alone (tools/producer_bench.cpp: files -> slab rows, rows discarded; no GPU): sites/s by host threads"
P1=""
for t in 1 2 4 8 16 32 64 96; do
  L=$($W/pbench $t $BF); echo "$L" >> $OUT
  R=$(echo "$L" | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['sites_per_s'])")
  [ -z "$P1" ] && P1=$R
  say "$(python3 -c "
q = $QUOTA
cpus = min($t, q) if q > 0 else $t
print('   --thread %3d: %9.0f sites/s   speed-up %5.1f   parallel efficiency %.2f (of %g CPUs)' % ($t, $R, $R / $P1, $R / $P1 / cpus, cpus))")"
done
say "-- bv_call end to end (producer -> engine -> emitter), one engine; engine idle = 1 - engine seconds / elapsed"
for t in 1 4 16 32 64; do
  $CALL --batchfiles $BF --output-vcf $W/b_t$t.vcf --output-cvg $W/b_t$t.cvg --thread $t --timing $W/b.json 2>&1 | tail -2 | tee -a $OUT
  cat $W/b.json >> $OUT
  say "$(python3 -c "import json; d = json.load(open('$W/b.json')); print('   --thread %3d: %9.0f sites/s end to end, engine idle %.2f' % ($t, d['sites_per_s'], 1 - d['engine_s'] / d['total_s']))")"
  [ $t != 1 ] && cmp $W/b_t$t.vcf $W/b_t1.vcf && cmp $W/b_t$t.cvg $W/b_t1.cvg && say "   outputs with --thread $t: byte-identical to one thread"
done
rm -rf $W
