export HSA_ENABLE_IPC_MODE_LEGACY=0
F="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
ms() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'])" $1 "$2"; }
python bench.py $F > gpurun_out/ab1.json 2>/dev/null; ms gpurun_out/ab1.json "plain"
OMP_NUM_THREADS=1 python bench.py $F > gpurun_out/ab2.json 2>/dev/null; ms gpurun_out/ab2.json "plain OMP=1"
TTR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py $F > gpurun_out/ab3.json 2>/dev/null; ms gpurun_out/ab3.json "torchrun forced dist, gather end"
TTR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py $F --gather none > gpurun_out/ab4.json 2>/dev/null; ms gpurun_out/ab4.json "torchrun forced dist, gather none"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py $F > gpurun_out/ab5.json 2>/dev/null; ms gpurun_out/ab5.json "torchrun, single rank, not forced"
python bench.py $F > gpurun_out/ab6.json 2>/dev/null; ms gpurun_out/ab6.json "plain again"
