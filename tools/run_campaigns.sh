#!/bin/bash
# the differential campaigns of a round against the real reference: tools/run_campaigns.sh <tag> [rounds per campaign] [seed offset]
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
TAG=${1:-r5}; R=${2:-30}; OFF=${3:-0}; OUT=gpurun_out/${TAG}_diff_campaign.txt; : > $OUT
[ "$OFF" != 0 ] && echo "# every CAMPAIGN_SEED below + $OFF" >> $OUT
run() { echo "### $1" >> $OUT; shift; local a=(); for kv in "$@"; do case $kv in CAMPAIGN_SEED=*) a+=("CAMPAIGN_SEED=$(( ${kv#CAMPAIGN_SEED=} + OFF ))");; *) a+=("$kv");; esac; done; env "${a[@]}" timeout 1500 python3 tools/diff_campaign.py $R 2>&1 | grep -E "TOTAL|tie-excused site [0-9]|MISMATCH|mismatching fields [1-9]" >> $OUT; }
run "fused short-row kernel: rows of 4,097-49,152 samples, seed 61" CAMPAIGN_FUSED=1 CAMPAIGN_SEED=61
run "fused short-row kernel, seed 62" CAMPAIGN_FUSED=1 CAMPAIGN_SEED=62
run "fused short-row kernel, chained launches incl. pop-groups, seed 63" CAMPAIGN_FUSED=1 CAMPAIGN_CHAIN=1 CAMPAIGN_SEED=63
run "fused short-row kernel, two workgroups (BV_FLAG_GRID_LIMIT), seed 64" CAMPAIGN_FUSED=1 CAMPAIGN_FLAGS=$((2 << 16)) CAMPAIGN_SEED=64
run "fused short-row kernel, ONE workgroup (every queue overflows), seed 65" CAMPAIGN_FUSED=1 CAMPAIGN_FLAGS=$((1 << 16)) CAMPAIGN_SEED=65
run "fused short-row kernel, three workgroups, chained, seed 66" CAMPAIGN_FUSED=1 CAMPAIGN_CHAIN=1 CAMPAIGN_FLAGS=$((3 << 16)) CAMPAIGN_SEED=66
run "fused short-row kernel, two workgroups, no rank planes, seed 67" CAMPAIGN_FUSED=1 CAMPAIGN_NORANKS=1 CAMPAIGN_FLAGS=$((2 << 16)) CAMPAIGN_SEED=67
run "default shapes, seed 11" CAMPAIGN_SEED=11
run "default shapes, seed 12" CAMPAIGN_SEED=12
run "default shapes, seed 13" CAMPAIGN_SEED=13
run "shallow rows (ties), seed 21" CAMPAIGN_SHALLOW=1 CAMPAIGN_SEED=21
run "shallow rows (ties), seed 22" CAMPAIGN_SHALLOW=1 CAMPAIGN_SEED=22
run "chained launches incl. pop-groups, seed 31" CAMPAIGN_CHAIN=1 CAMPAIGN_SEED=31
run "chained launches incl. pop-groups, seed 32" CAMPAIGN_CHAIN=1 CAMPAIGN_SEED=32
run "chained + shallow, seed 33" CAMPAIGN_CHAIN=1 CAMPAIGN_SHALLOW=1 CAMPAIGN_SEED=33
run "three-launch short-row form BV_FLAG_SHORT_ROW_FORM(9), seed 41" CAMPAIGN_FLAGS=$((9 << 12)) CAMPAIGN_SEED=41
run "wave solver only (BV_FLAG_WAVE_SOLVER), seed 51" CAMPAIGN_FLAGS=16 CAMPAIGN_SEED=51
run "tile jobs, per-site tallies (BV_FLAG_TILE_STATE): shallow replay from the cell lists, groups, seed 71" CAMPAIGN_TILES=1 CAMPAIGN_FLAGS=8 CAMPAIGN_SEED=71
run "tile jobs, per-site tallies, seed 72" CAMPAIGN_TILES=1 CAMPAIGN_FLAGS=8 CAMPAIGN_SEED=72
run "tile jobs, joined rows, seed 73" CAMPAIGN_TILES=1 CAMPAIGN_SEED=73
cat $OUT
