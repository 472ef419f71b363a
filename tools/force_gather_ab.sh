set -e -o pipefail
mkdir -p gpurun_out/fg
for r in 1 2; do
  for f in 0 1; do
    RECO_FORCE_GATHER=$f timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$r bench.py --gpus 1 --no-cpu-baseline --no-kernels --no-split > gpurun_out/fg/force${f}_$r.json 2> gpurun_out/fg/force${f}_$r.err
    echo "force=$f round $r done"
  done
done
