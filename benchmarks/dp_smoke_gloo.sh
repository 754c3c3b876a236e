# Functional smoke run of the N = 2 data-parallel control flow on a ONE-GPU box: two ranks on GPU 0, gloo carrying the collectives
# (RCCL refuses two ranks on one device).  Not a measurement.  usage: bash benchmarks/dp_smoke_gloo.sh [bf16|fp32]
DT=${1:-fp32}
REFTR_DIST_BACKEND=gloo REFTR_BENCH_ONE_DEVICE=1 REFTR_DDP_DTYPE=$DT timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-roofline 2>&1 | tail -4
