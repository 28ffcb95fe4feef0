cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_commit_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "== K=4"; timeout 200 python tools/lde_throughput.py 20x64 20x78 19x314 18x114 2>&1 | grep "2^"
echo "== K=1"; LURKHIP_LIB_PATH=$GRAFT_REPO_ROOT/lurk_amd/liblurkhip_mx0.so timeout 200 python tools/lde_throughput.py 20x64 20x78 19x314 18x114 2>&1 | grep "2^"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-pipeline > gpurun_out/b5.log 2>&1; python tools/show_bench.py gpurun_out/b5.log 2>/dev/null | head -20
