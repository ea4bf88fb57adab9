mkdir -p gpurun_out/r06
for i in 1 2; do
TTR_KNOBS=5=4 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_nw8_$i.txt 2> gpurun_out/r06/bench_nw8_$i.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_def_$i.txt 2> gpurun_out/r06/bench_def_$i.err
done
