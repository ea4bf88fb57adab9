mkdir -p gpurun_out/r06
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_prio_$i.txt 2> gpurun_out/r06/bench_prio_$i.err
TTR_LIB_PATH=tntorch_amd/libttround_base.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_base_$i.txt 2> gpurun_out/r06/bench_base_$i.err
done
python tools/probes/qr_metric_stamps.py 4096 > gpurun_out/r06/metric_stamps_prio.txt 2>&1
