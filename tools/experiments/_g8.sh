cd $GRAFT_REPO_ROOT
timeout 1500 tools/host_pipeline.sh r5 24000 2>&1 | grep -A60 "== (b)"
