#!/bin/bash
# Host pipeline end to end (bv_call): stage seconds and sites/s, the way the reference prints its phases.
#   tools/host_pipeline.sh <tag>        -> gpurun_out/host_pipeline_<tag>.txt  (+ .json lines)
# (a) the reference's 100-BAM test set (copied to tests/golden/_local/bam100 by `tools/host_pipeline.sh stage` in the build
#     container; all-N surrogate FASTA as in tools/real_data_campaign.py), BAM -> pileup -> engine -> VCF/CVG;
# (b) synthetic batchfiles, 10,000 samples in files of 200 (the reference's --batch-count), the byte-level reader on 1 / 4 / 16 host threads.
cd "$(dirname "$0")/.."; ROOT=$PWD
if [ "$1" = "stage" ]; then
  mkdir -p tests/golden/_local/bam100
  cp /root/reference/tests/data/140k_thalassemia_brca_bam/bam100/*.bam* tests/golden/_local/bam100/ 2>/dev/null
  cp /root/reference/tests/data/140k_thalassemia_brca_bam/sample_group.info tests/golden/_local/bam100/
  ls tests/golden/_local/bam100 | wc -l
  exit 0
fi
TAG=${1:-r3}; SITES=${2:-1500}; shift; shift
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
mkdir -p $W/bf; $W/gen $W/bf 10000 200 $SITES 0.08 7
BF=$(ls $W/bf/*.gz | paste -sd,)
say "== (b) synthetic batchfiles: 10000 samples in $(ls $W/bf | wc -l) files, $SITES sites, $(du -sh $W/bf | cut -f1) gzip"
say "-- one host thread"
$CALL --batchfiles $BF --output-vcf $W/b_fast.vcf --output-cvg $W/b_fast.cvg --timing $W/b.json 2>&1 | tail -2 | tee -a $OUT
cat $W/b.json >> $OUT
for t in 4 16; do
  say "-- --thread $t (files read and sites parsed in blocks by $t threads, lines formatted by $t threads)"
  $CALL --batchfiles $BF --output-vcf $W/b_t.vcf --output-cvg $W/b_t.cvg --thread $t --timing $W/b.json 2>&1 | tail -2 | tee -a $OUT
  cat $W/b.json >> $OUT
  cmp $W/b_t.vcf $W/b_fast.vcf && cmp $W/b_t.cvg $W/b_fast.cvg && say "outputs with --thread $t: byte-identical to one thread"
done
rm -rf $W
