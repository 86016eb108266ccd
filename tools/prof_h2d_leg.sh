set -u
out=$PWD/gpurun_out/${1:-r06ap}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH_ONLY_LEGS=pcie_inclusive timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof -o p --output-format csv -- python /root/repo/bench.py --steps 100 --warmup 5 --no-traffic --no-cpu-baseline --no-roofline > $out/log.txt 2>&1
ls $out/prof/* | head; for f in $(find $out/prof -name "*kernel_stats.csv"); do head -8 $f | cut -c1-160; done
for f in $(find $out/prof -name "*memory_copy_stats.csv"); do cat $f | cut -c1-200; done
