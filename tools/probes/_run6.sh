mkdir -p gpurun_out/r06
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_pre_$i.txt 2> gpurun_out/r06/bench_pre_$i.err
TTR_KNOBS=17=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_nopre_$i.txt 2> gpurun_out/r06/bench_nopre_$i.err
done
python tools/probes/qr_metric_stamps.py 4096 > gpurun_out/r06/metric_stamps_pre.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r06/gputest6.txt 2>&1
tail -3 gpurun_out/r06/gputest6.txt
