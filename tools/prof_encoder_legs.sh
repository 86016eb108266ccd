# rocprofv3 kernel tables of the 512-wide encoder steps on the final tree, ONE batch in flight (clean kernel durations), the encoder told
# about the batches the bench keeps in flight (BENCH_ENC_IN_FLIGHT) so that the launch sequence is the timed one
set -u
out=$PWD/gpurun_out/${1:-r06av}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "large 64 2" "ebf 32 3" "large 32 2"; do
  set -- $cfg
  BENCH_ENC_IN_FLIGHT=$3 timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$1_$2 -o p --output-format csv -- python /root/repo/bench.py --model $1 --batch $2 --steps 20 --warmup 3 --quick --no-cpu-baseline --no-traffic --no-roofline --in-flight 1 > $out/log_$1_$2.txt 2>&1
  f=$(find $out/prof_$1_$2 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/bench_$1_b$2_kernel_stats.csv && echo "== $1 B=$2" && head -16 $out/bench_$1_b$2_kernel_stats.csv | cut -c1-170
  rm -rf $out/prof_$1_$2
done
