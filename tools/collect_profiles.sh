#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 evidence for every shipped kernel.
#   tools/collect_profiles.sh <tag>          e.g. r2            (ONLY="name name ..." re-collects just those configurations)
# For each configuration below: one un-profiled bench run (the JSON line), one `--kernel-trace --stats` run and two PMC
# runs (FETCH_SIZE, WRITE_SIZE -- separate passes, no tracing beside them), all of the same bench command.
# Back in the container:  python tools/summarize_profiles.py <tag>
TAG=${1:-r6}
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
O=$ROOT/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
run_cfg() {  # name, bench args
  local name=$1; shift
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  echo "== $name: $*" >&2
  echo "$*" > $O/$name.args
  timeout 600 python3 bench.py --no-cpu-baseline "$@" > $O/$name.bench.json 2> $O/$name.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name.stats -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > /dev/null 2>> $O/$name.err
  # per-dispatch rows of our kernels only (the summary groups calls by dispatch size); the full trace is not kept
  for f in $(find $O/$name.stats -name '*kernel_trace.csv'); do
    python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if r["Kernel_Name"].replace("void ", "").startswith("bv_") and "synth" not in r["Kernel_Name"]]
if keep:
    cols = [c for c in ("Kernel_Name", "Grid_Size_X", "Grid_Size", "Workgroup_Size_X", "Workgroup_Size", "LDS_Block_Size", "Start_Timestamp", "End_Timestamp") if c in keep[0]]
    w = csv.DictWriter(open(sys.argv[1].replace("kernel_trace.csv", "bv_dispatches.csv"), "w"), fieldnames=cols)
    w.writeheader()
    for r in keep:
        w.writerow({c: r[c] for c in cols})
PY
  done
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$name.fetch -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>> $O/$name.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$name.write -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>> $O/$name.err
  # keep only the CSVs that the summary reads (the merge back is capped at 64 MiB)
  find $O/$name.stats $O/$name.fetch $O/$name.write -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name '*bv_dispatches.csv' -delete 2>/dev/null
}
sq_cfg() {  # name, bench args: SQ instruction-mix / wait counters of every kernel, three passes
  local name=$1; shift
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name.sq "; then return; fi
  local i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/$name.sq$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>> $O/$name.err
    find $O/$name.sq$i -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
    i=$((i+1))
  done
}
run_cfg n100k                                               # headline (tagged ranks): bv_pass1_kernel<3,1,false,2>, bv_pass2_kernel<256,true,false,true,false,true> (TAG: 3 B/cell)
run_cfg n100k_plain --rank-layout plain                     # the same rows with plain ranks: bv_pass2_kernel<256,true,false> re-reads the call plane (4 B/cell)
run_cfg n100k_groups2 --groups 2 --batch-sites 65536        # bv_pass2_kernel<256,true,true,false> + bv_p2g_solve16_kernel on long rows
run_cfg n1M --samples 1000000 --batch-sites 16384 --steps 8 # the same kernels at 1 M samples
run_cfg n10k --samples 10000 --batch-sites 100000           # configs[1] (tagged ranks): bv_p1s_fused_kernel<true> (pass 1 + the variant sites' pass-2 rows in one persistent kernel)
run_cfg n10k_plain --samples 10000 --batch-sites 100000 --rank-layout plain   # ... with plain ranks
# dense rows (round 6): coverage 1.0 -- the dominant-value count in the mapq tallies, the bank swizzle of dense long rows
run_cfg n100k_cov1 --coverage 1.0 --batch-sites 65536 --steps 10
run_cfg n10k_cov1 --samples 10000 --batch-sites 100000 --coverage 1.0
run_cfg n10k_p2sep --samples 10000 --batch-sites 100000 --flags 40960   # BV_FLAG_SHORT_ROW_FORM(10): pass 2 a launch of its own: bv_p1s_fused_kernel<false> + bv_pass2_dma_kernel
run_cfg n10k_three_launches --samples 10000 --batch-sites 100000 --flags 36864  # BV_FLAG_SHORT_ROW_FORM(9): the kernels rows of <= 4,096 samples take: bv_p1s_stream_kernel, bv_p1s_solve16_kernel, bv_pass2_dma_kernel
run_cfg n10k_lanes2 --samples 10000 --batch-sites 100000 --lanes 2   # the same through the engine's two lanes (BV_FLAG_LANES)
run_cfg n10k_524k --samples 10000 --batch-sites 524288
# the fused kernel's HBM writes by launch size: scratch memory of the persistent grid is the fixed part
run_cfg n10k_2048 --samples 10000 --batch-sites 2048 --steps 40
run_cfg n10k_8192 --samples 10000 --batch-sites 8192 --steps 30
run_cfg n10k_32768 --samples 10000 --batch-sites 32768 --steps 20
run_cfg n10k_groups2 --samples 10000 --batch-sites 100000 --groups 2   # short rows with pop-groups: bv_p2g_stream_kernel + bv_p2g_solve16/hard behind the fused kernel
run_cfg n10k_groups1 --samples 10000 --batch-sites 100000 --groups 1
run_cfg n10k_noranks --samples 10000 --batch-sites 100000 --groups 2 --no-rank-planes   # groups without rank planes: bv_p2g_stream_kernel alone
run_cfg n10k_groups8 --samples 10000 --batch-sites 100000 --groups 8   # more than 7 groups: bv_pass2_kernel<256,false,true,false> (group tallies only: the fused kernel streams the rank-sum rows) + the group solve kernels
run_cfg n10k_groups16 --samples 10000 --batch-sites 100000 --groups 16 # bv_p2g_solve_small_kernel<8>: eight items per wave
run_cfg n10k_groups32 --samples 10000 --batch-sites 100000 --groups 32 # one round of groups: every group of the cohort is shallow (<= 64 covered samples)
run_cfg n10k_groups64 --samples 10000 --batch-sites 100000 --groups 64 # two rounds of 32 groups (BV_GROUPS_PER_ROUND): pass 2 runs twice, the rank sums once
run_cfg n100k_chain16 --batch-sites 8192 --chain 16                    # small batches chained: bv_pass1_kernel<3,1,true>
run_cfg n100k_8192 --batch-sites 8192 --steps 20                       # ... and one launch per small batch
run_cfg n10k_chain16 --samples 10000 --batch-sites 8192 --chain 16     # short rows chained: bv_p1s_fused_kernel<true> over the queue, bv_chain_* kernels
run_cfg tiles_joined_1M --samples 1000000 --batch-sites 8192 --tile-sites 8192 --with-tile-mode --steps 2 --warmup 1   # bv_tile_join_rows_kernel (bv_engine_tiles_add_many), bv_tile_scatter_kernel (tile by tile)
run_cfg tiles_state_100k --samples 100000 --batch-sites 16384 --tile-sites 16384 --with-tile-mode --flags 8 --steps 2 --warmup 1   # bv_tile_tally_kernel, bv_tile_finish_kernel
run_cfg tile_job_1M --tile-job --tile-sites 8192 --steps 2 --warmup 1  # BASELINE configs[4] shape as the timed workload: host tiles -> bv_tile_scatter_kernel -> both passes
sq_cfg n10k --samples 10000 --batch-sites 100000
sq_cfg n100k
ls $O | head -80 >&2
du -sh $O >&2
