set -x
mkdir -p gpurun_out/r06
cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/push_membench push_membench.hip && cd ../.. && /tmp/push_membench 4096 > gpurun_out/r06/push_membench.txt 2>&1
python tools/probes/qr_push_ab.py 16 0,24,45,60 4096 > gpurun_out/r06/stagger_ab.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r06/bench_base.txt 2> gpurun_out/r06/bench_base.err
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gputest1.txt 2>&1
tail -5 gpurun_out/r06/gputest1.txt
