cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['sample_launch_ms'], d['config']['survivors'])"
mkdir -p /tmp/fw && cd /tmp/fw && for i in 1 2 3; do python $GRAFT_REPO_ROOT/filter.py --dataset ppa --model adamic_ogb --checkpoint "ppa_adamic_ogb||0|0.pt" --synthetic --keep_top 4000000 2>&1 | grep -E "scored in|threshold scan"; done
