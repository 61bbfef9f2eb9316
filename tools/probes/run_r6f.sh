cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
timeout 900 python -m pytest tests/test_draft_gpu.py -x -q -m gpu > gpurun_out/r6f/tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6f/tests.txt
bash tools/profile_draft.sh r6f 2>&1 | tail -30
