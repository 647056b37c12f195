cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scan.py -x -q 2>&1 | tail -5
timeout 300 python tools/scan_bench.py --reps 3 2>&1 | grep -v amdgpu.ids | tail -8
timeout 300 python tools/scan_stamps.py 2>&1 | grep -v amdgpu.ids | tail -10
