set -x
mkdir -p gpurun_out/r06
python tools/probes/qr_metric_stamps.py 4096 > gpurun_out/r06/metric_stamps_v4.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v4.txt 2> gpurun_out/r06/bench_v4.err
TTR_LIB_PATH=tntorch_amd/libttround_pushv1.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v1c.txt 2> gpurun_out/r06/bench_v1c.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v4b.txt 2> gpurun_out/r06/bench_v4b.err
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r06/gputest4.txt 2>&1
tail -5 gpurun_out/r06/gputest4.txt
