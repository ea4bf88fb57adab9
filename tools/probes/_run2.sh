set -x
mkdir -p gpurun_out/r06
cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/push_membench push_membench.hip && cd ../.. && /tmp/push_membench 4096 > gpurun_out/r06/push_membench2.txt 2>&1
python tools/probes/qr_push_ab.py 16 0 4096 > gpurun_out/r06/push_v2.txt 2>&1
TTR_LIB_PATH=tntorch_amd/libttround_pushv1.so python tools/probes/qr_push_ab.py 16 0 4096 > gpurun_out/r06/push_v1.txt 2>&1
python tools/probes/qr_push_ab.py 16 0 4096 >> gpurun_out/r06/push_v2.txt 2>&1
TTR_LIB_PATH=tntorch_amd/libttround_pushv1.so python tools/probes/qr_push_ab.py 16 0 4096 >> gpurun_out/r06/push_v1.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v2.txt 2> gpurun_out/r06/bench_v2.err
TTR_LIB_PATH=tntorch_amd/libttround_pushv1.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v1.txt 2> gpurun_out/r06/bench_v1.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_v2b.txt 2> gpurun_out/r06/bench_v2b.err
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r06/gputest2.txt 2>&1
tail -5 gpurun_out/r06/gputest2.txt
