set -x
mkdir -p gpurun_out/r06
for lib in hip nozskip nomma pushv1; do
  TTR_LIB_PATH=tntorch_amd/libttround_$lib.so python tools/probes/qr_push_ab.py 16 0 4096 > gpurun_out/r06/p3_${lib}_4096.txt 2>&1
  TTR_LIB_PATH=tntorch_amd/libttround_$lib.so python tools/probes/qr_push_ab.py 16 0 64 > gpurun_out/r06/p3_${lib}_64.txt 2>&1
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v3.txt 2> gpurun_out/r06/bench_v3.err
TTR_LIB_PATH=tntorch_amd/libttround_pushv1.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v1b.txt 2> gpurun_out/r06/bench_v1b.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v3b.txt 2> gpurun_out/r06/bench_v3b.err
