mkdir -p gpurun_out/r05d
python -m pytest tests/test_prover_gpu.py tests/test_cpu_step_gpu.py tests/test_air_gpu.py tests/test_workloads_gpu.py -q -x > gpurun_out/r05d/tests.log 2>&1; tail -3 gpurun_out/r05d/tests.log
python bench.py --no-cpu-baseline --no-host-pipeline > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err
LURKHIP_QUOTIENT_READ_NEXT_SUM=1 python bench.py --no-cpu-baseline --no-host-pipeline > gpurun_out/r05d/bench_readnext.json 2> gpurun_out/r05d/bench_readnext.err
python tools/show_bench.py gpurun_out/r05d/bench.json gpurun_out/r05d/bench_readnext.json | head -20
