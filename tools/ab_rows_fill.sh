# 512-wide encoders, row-block launches (64-row workgroups) against the per-operator sequence (ESPNET_AMD_NO_FFN_ROWS=1) with 1 / 2 / 3
# batches in flight (bench.py tells the library through EM_ENC_IN_FLIGHT; the fill rule is 72 % / n of the CUs)
set -u
out=gpurun_out/${1:-r06af}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ebranchformer.py tests/test_gpu_e2e.py -q -x -k "ebranchformer or large or ebf" 2>&1 | tail -12 | tee $out/pytest.txt
for M in ebf large; do
 for B in 32; do
  for IF in 1 2 3; do
   for NOROWS in 0 1; do
    echo "== $M B=$B in_flight=$IF no_ffn_rows=$NOROWS" | tee -a $out/ab.txt
    if [ $NOROWS = 1 ]; then export ESPNET_AMD_NO_FFN_ROWS=1; else unset ESPNET_AMD_NO_FFN_ROWS; fi
    timeout 200 python bench.py --model $M --batch $B --steps 100 --warmup 10 --quick --no-cpu-baseline --no-traffic --no-roofline --in-flight $IF 2>$out/err_${M}_${B}_${IF}_${NOROWS}.txt | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'))" | tee -a $out/ab.txt
   done
  done
 done
done
unset ESPNET_AMD_NO_FFN_ROWS
echo "== large B=64 in_flight=2" | tee -a $out/ab.txt
timeout 200 python bench.py --model large --batch 64 --steps 60 --warmup 10 --quick --no-cpu-baseline --no-traffic --no-roofline --in-flight 2 2>$out/err_large64.txt | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'))" | tee -a $out/ab.txt
