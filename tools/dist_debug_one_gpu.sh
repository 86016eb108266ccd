# developer check of bench.py's multi-rank control flow on a ONE-GPU box (both ranks on cuda:0, collectives through gloo on host copies): not a measurement
set -u
out=gpurun_out/${1:-r06as}; mkdir -p $out
for WL in greedy beam; do
  echo "== --gpus 2 --workload $WL --dist-debug-one-gpu" | tee -a $out/dist.txt
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 40 --warmup 5 --workload $WL --batch 16 --dist-debug-one-gpu --quick --no-cpu-baseline --no-traffic --no-roofline 2>$out/err_$WL.txt | tail -1 | cut -c1-600 | tee -a $out/dist.txt
  tail -3 $out/err_$WL.txt | grep -v amdgpu.ids
done
