cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or tallies" 2>&1 | tail -15 | cut -c1-600
timeout 1500 tools/host_pipeline.sh r5 12000 2>&1 | tail -40
cp gpurun_out/host_pipeline_r5.txt gpurun_out/host_pipeline_r5_copy.txt
