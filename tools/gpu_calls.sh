#!/bin/bash
# The GPU-box calls of rounds 3 and 4 in one place (each was one `gpurun -- bash tools/gpu_calls.sh <what> <tag>`; output under
# gpurun_out/<tag>/, the summaries that are quoted in DESIGN.md copied to profiles/<tag>_*).
#   full         the whole `pytest -m gpu` suite, smoke, the default bench line, rocprofv3 table of the greedy step
#   bench        smoke, the default bench line (roofline + traffic + cpu baseline + beam + stream legs), rocprofv3 table
#   sub2         fused subsampling kernel: cycle stamps of one workgroup, the conv tests, per-kernel stats, quick bench line
#   search-stats per-kernel stats of the beam-search label step
#   search-pmc   ... plus SQ wave-cycle breakdown and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes)
#   sa-ab        decoder self-attention: waves per row x rows per workgroup (ESPNET_AMD_SA_SPLIT / _GROUP), label-step A/B
#   attn-stamps  relpos_attn2 cycle stamps (EM_ATTN2_STAMPS)
#   greedy-pmc   SQ counters (wave cycles, waits, MFMA busy, LDS bank conflicts) of the greedy step's main kernels
#   ffn-rows     round 4: row-block launches of the 512-wide model - kernel + large e2e tests, in-call A/B (ESPNET_AMD_NO_FFN_ROWS), stamps, table
#   ffn-dbg      ... where ffn_rows_kernel's time goes: developer builds (no weight requests / no LDS operand reads / both) and cycle stamps
#   r05a|r05b    round 5: new parity tests (each file under its own limit), block / frontend A/B, fine stamps, launch order of one step
#   r05c         round 5: block.hip stream across stage boundaries + per-row conv requests against lib_prev / lib_v64
#   attn-stamps5 relpos_attn2 cycle stamps, small model
#   search640, sweep640   label step at configs[3]'s per-GPU shape: kernel table; sweep of the dispatch switches (B64=16 for 160 rows)
#   tree-sa      round 5: decoder self-attention over the union of a beam's ancestors: tests, in-call A/B, 640-row kernel table
#   stream-ab    round 5: helper workgroups on a partly filled chip + block attention on MFMA: streaming A/B, suite, smoke, bench, 32-stream table
set -u
what=${1:-bench}; tag=${2:-r03}; out=$PWD/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
R=$PWD
stats() {  # <dir> <cmd...>: rocprofv3 --kernel-trace --stats of a command, summary kept, trace dropped
  local d=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$d" -o s --output-format csv -- "$@" > "$d.log" 2>&1 < /dev/null)
  find "$d" -name "*_kernel_trace.csv" -delete 2>/dev/null
  local f; f=$(find "$d" -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then head -16 "$f" | cut -c1-170; else echo "no stats file"; tail -5 "$d.log"; fi
}
beam_line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])"; }
quick() { timeout 300 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps ${1:-600} --warmup 30 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms/step')"; }
case "$what" in
  r06t)  # round 6: split FFN of the streaming row-block launches (EmBlockArgs.ffn_split): streaming + block tests, the sweep
         # over streams x shares, in-call A/B (ESPNET_AMD_STREAM_FFN_SPLIT=1 + _FUSED_MIN=8 is round 5's form), 1 / 32 / 128 streams
    echo "== tests"
    (timeout 900 python -m pytest -q -x tests/test_gpu_streaming.py tests/test_gpu_online_search.py tests/test_gpu_block.py 2>&1 | tail -6) | tee "$out/pytest.txt"
    echo "== sweep"
    timeout 600 python tools/experiments/stream_split_sweep.py 2>&1 | grep streams | tee "$out/split_sweep.txt"
    echo "== A/B"
    for v in new r05 new r05; do
      if [ $v = r05 ]; then export ESPNET_AMD_STREAM_FFN_SPLIT=1 ESPNET_AMD_STREAM_FUSED_MIN=8; else unset ESPNET_AMD_STREAM_FFN_SPLIT ESPNET_AMD_STREAM_FUSED_MIN; fi
      echo -n "$v: "
      timeout 600 python bench.py --workload stream --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s one stream, call median', d['config']['call_latency_ms_median'], 'ms; batch32', d.get('batch32',{}).get('value'), 'tick', d.get('batch32',{}).get('tick_latency_ms_median'), 'batch128', d.get('batch128',{}).get('value'))"
    done 2>&1 | tee "$out/ab_ffn_split.txt"
    unset ESPNET_AMD_STREAM_FFN_SPLIT ESPNET_AMD_STREAM_FUSED_MIN ;;
  r06u)  # round 6: kernel tables of the streaming call with the split FFN: one stream (hipGraph step) and the 32-stream tick
    echo "== one stream"; stats "$out/stream1_stats" python "$R/tools/stream_ab.py" one
    f=$(find "$out/stream1_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/stream_one_kernel_stats.csv"
    echo "== 32 streams"; stats "$out/stream32_stats" python "$R/tools/stream_ab.py" batch
    f=$(find "$out/stream32_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/stream_batch32_kernel_stats.csv" ;;
  r06a)  # round 6: block<ATT|C> (attention + the C part in one launch): kernel tests, e2e parity of the small model, in-call A/B
         # against the two-launch form (ESPNET_AMD_SPLIT_ATT=1), kernel table
    echo "== box state"; BOX_STATE_OUT="$out/box_state" bash tools/box_state.sh 2>&1 | tail -4 | tee "$out/box_state.txt"
    echo "== kernel tests"
    (timeout 600 python -m pytest -q -x tests/test_gpu_block.py 2>&1 | tail -6) | tee "$out/pytest_block.txt"
    echo "== e2e parity (small model paths)"
    (timeout 900 python -m pytest -q -x tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -k "not beam" 2>&1 | tail -6) | tee "$out/pytest_e2e.txt"
    echo "== A/B: split=1 is attention2 + block<C> (rounds 2-5), split=0 is block<ATT|C>"
    for v in 0 1 0 1; do
      if [ $v = 1 ]; then export ESPNET_AMD_SPLIT_ATT=1; else unset ESPNET_AMD_SPLIT_ATT; fi
      echo -n "split_att=$v: "; quick 600
    done 2>&1 | tee "$out/ab_att_c.txt"
    unset ESPNET_AMD_SPLIT_ATT
    echo "== stamps"; EM_BLOCK_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep -E "block<65>" | tail -3 | cut -c1-600 | tee "$out/block_stamps.txt"
    echo "== kernel table"
    stats "$out/prof_greedy" python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 100 --warmup 10 | cut -c1-200
    f=$(find "$out/prof_greedy" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_small_b32_kernel_stats.csv" ;;
  r05a|r05b)  # round 5: kernel tests of the changed kernels, LayerNorm hand-over with the weight requests dealt into it (A/B against
            # lib_v32 = requests in one cluster), frontend v2 (A/B against ESPNET_AMD_FRONTEND_V1), fine stamps, kernel table, launch
            # order of one step (where do the ~7 copyBuffer launches per step come from?), then the new parity tests, each
            # file under its own limit (r05a lost 25 GPU-minutes to oracle legs on an unbounded thread count)
    echo "== kernel tests of the changed kernels"
    (timeout 300 python -m pytest -q -x tests/test_gpu_block.py tests/test_gpu_kernels.py -k "block or frontend or mvn" 2>&1 | tail -4) | tee "$out/pytest_kernels.txt"
    echo "== A/B: LayerNorm hand-over (v32 = weight requests in one cluster in front of the LayerNorm)"
    for v in new v32 new v32; do
      if [ $v = new ]; then unset ESPNET_AMD_LIB; else export ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_$v.so; fi
      echo -n "block $v: "; quick 600
    done 2>&1 | tee "$out/ab_block.txt"
    unset ESPNET_AMD_LIB
    echo "== A/B: frontend"
    for v in 0 1 0 1; do
      if [ $v = 1 ]; then export ESPNET_AMD_FRONTEND_V1=1; else unset ESPNET_AMD_FRONTEND_V1; fi
      echo -n "frontend_v1=$v: "; quick 600
    done 2>&1 | tee "$out/ab_frontend.txt"
    unset ESPNET_AMD_FRONTEND_V1
    echo "== fine stamps"; ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_fine.so EM_BLOCK_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep -E "block<(1|6)>" | tail -8 | cut -c1-900 | tee "$out/block_stamps_fine.txt"
    echo "== kernel stats + launch order of one step"
    d="$out/prof_greedy"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$d" -o s --output-format csv -- python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 20 --warmup 3 > "$d.log" 2>&1 < /dev/null)
    f=$(find "$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150
    t=$(find "$d" -name "*kernel_trace.csv" | head -1)
    [ -n "$t" ] && python - "$t" <<'PY' | tee "$out/launch_order.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "frontend_logmel" in n]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
PY
    find "$d" -name "*_kernel_trace.csv" -delete 2>/dev/null
    echo "== new parity tests"
    for sel in "tests/test_gpu_e2e.py -k large_peaked_or_peaked_posteriors_or_bfloat16_within_or_large_rows" \
               "tests/test_gpu_scorer_interface.py -k 640_rows" \
               "tests/test_gpu_search.py -k dec_self_attention_or_long_memory_or_lnq" \
               "tests/test_gpu_fullsize.py -k b16_rows_bf16_vs_oracle" \
               "tests/test_gpu_fullsize.py -k b64_rows_bf16_vs_oracle"; do
      sel=${sel//_or_/ or }
      f=${sel%% *}; k=${sel#* -k }
      (time timeout 420 python -m pytest -q -s "$f" -k "$k" > "$out/pytest_one.txt" 2>&1; grep -E "^\[|passed|failed|Error|assert " "$out/pytest_one.txt" | cut -c1-420 | tail -24) 2>&1 | tee -a "$out/pytest_parity.txt"
    done ;;
  r05c)     # round 5: the weight stream carried across stage boundaries + one request per row of the depthwise conv
            # (A/B: lib_prev = block.hip of the commit before, lib_v64 = new stream, conv requests in three clusters)
    echo "== tests of the changed kernels (block kernels, streaming layers on them, end-to-end bf16 goldens)"
    (timeout 600 python -m pytest -q -x tests/test_gpu_block.py tests/test_gpu_streaming.py tests/test_gpu_e2e.py -k "block or fused or stream or bfloat16_within or peaked or midmargin or per_operator" 2>&1 | tail -4) | tee "$out/pytest_kernels.txt"
    echo "== A/B"
    for v in new prev v64 new prev v64; do
      if [ $v = new ]; then unset ESPNET_AMD_LIB; else export ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_$v.so; fi
      echo -n "block $v: "; quick 600
    done 2>&1 | tee "$out/ab_block.txt"
    unset ESPNET_AMD_LIB
    echo "== fine stamps"; ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_fine.so EM_BLOCK_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep -E "block<(1|6)>" | tail -8 | cut -c1-900 | tee "$out/block_stamps_fine.txt"
    echo "== kernel stats"; stats "$out/prof_greedy" python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 50 --warmup 5
    echo "== stream batch32"; timeout 300 python - <<'PY' 2>/dev/null | tee "$out/stream_batch32.txt"
import json, bench
r = bench.run_stream_batch("bfloat16", 32, 3, 1)
print("batch32:", json.dumps({k: r[k] for k in r if k != "config"})[:300])
PY
    ;;
  attn-stamps5)  # relpos_attn2 cycle stamps of one wave (EM_ATTN2_STAMPS), small model
    EM_ATTN2_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep "attn2 stamps" | tail -6 | tee "$out/attn2_stamps.txt" ;;
  attn-stamps5)  # relpos_attn2 cycle stamps of one wave (EM_ATTN2_STAMPS), small model
    EM_ATTN2_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep "attn2 stamps" | tail -6 | tee "$out/attn2_stamps.txt" ;;
  r05g)     # attention: next tile's score / window MFMAs issued in front of the current tile's softmax chain (A/B: ESPNET_AMD_ATTN2_NOPIPE=1)
    echo "== tests"
    (timeout 600 python -m pytest -q -x tests/test_gpu_block.py tests/test_gpu_e2e.py tests/test_gpu_ebranchformer.py -k "attention2 or bfloat16_within or peaked or midmargin or large_rows or ebf" 2>&1 | tail -3) | tee "$out/pytest_kernels.txt"
    for v in 0 1; do
      if [ $v = 1 ]; then export ESPNET_AMD_ATTN2_NOPIPE=1; else unset ESPNET_AMD_ATTN2_NOPIPE; fi
      echo "== nopipe=$v small"; stats "$out/prof_small_$v" python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 50 --warmup 5 | grep -E "relpos" | sed 's/.*)",/  /'
      echo "== nopipe=$v large"; stats "$out/prof_large_$v" python "$R/bench.py" --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 20 --warmup 3 | grep -E "relpos" | sed 's/.*)",/  /'
    done 2>&1 | tee "$out/ab_attn_kernel_tables.txt"
    for f in small_0 small_1 large_0 large_1; do echo -n "$f: "; grep -h "relpos_attn" "$out"/prof_$f/*kernel_stats.csv | sed 's/.*)",//'; done | tee -a "$out/ab_attn_kernel_tables.txt"
    unset ESPNET_AMD_ATTN2_NOPIPE
    for v in 0 1 0 1; do
      if [ $v = 1 ]; then export ESPNET_AMD_ATTN2_NOPIPE=1; else unset ESPNET_AMD_ATTN2_NOPIPE; fi
      echo -n "nopipe=$v: "; quick 600
    done 2>&1 | tee "$out/ab_small.txt"
    unset ESPNET_AMD_ATTN2_NOPIPE ;;
  search640)  # label step at configs[3]'s per-GPU shape (64 x beam 10 = 640 rows): kernel table
    stats "$out/search640_stats" python "$R/bench.py" --workload beam --batch 64 --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic | cut -c1-200
    f=$(find "$out/search640_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-220 > "$out/search640_kernel_stats_head.txt" ;;
  sweep640)  # label step at 640 rows under the dispatch switches that were tuned at 160 rows
    B64=${B64:-64}
    bl() { timeout 200 python bench.py --workload beam --batch $B64 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s,', d['search']['ms_per_search_step'], 'ms per label step')"; }
    export B64
    for e in "" "ESPNET_AMD_SA_GROUP=1" "ESPNET_AMD_SA_GROUP=2" "ESPNET_AMD_SA_GROUP=3" "ESPNET_AMD_SA_GROUP=4" "ESPNET_AMD_MID_TILE=24" "ESPNET_AMD_MID_TILE=21" "ESPNET_AMD_MID_TILE=12" "ESPNET_AMD_SA_GROUP=4 ESPNET_AMD_MID_TILE=24" "ESPNET_AMD_SA_GROUP=2 ESPNET_AMD_MID_TILE=24" ""; do
      echo -n "[B=$B64 $e] "; env $e bash -c "$(declare -f bl); bl"
    done 2>&1 | tee "$out/sweep640.txt" ;;
  tree-sa)  # round 5: decoder self-attention over the union of a beam's ancestors (A/B: ESPNET_AMD_NO_SA_TREE=1)
    echo "== kernel test"; (timeout 300 python -m pytest -q -x tests/test_gpu_search.py -k "self_attention" 2>&1 | tail -5) | tee "$out/pytest_kernel.txt"
    echo "== search tests"; (timeout 600 python -m pytest -q -x tests/test_gpu_search.py tests/test_gpu_online_search.py -k "bf16 or peaked or batched or structure or very_short" 2>&1 | tail -4) | tee "$out/pytest_search.txt"
    bl() { timeout 200 python bench.py --workload beam --batch $1 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s,', d['search']['ms_per_search_step'], 'ms per label step')"; }
    for b in 64 16; do for v in 0 1 0 1; do
      if [ $v = 1 ]; then export ESPNET_AMD_NO_SA_TREE=1; else unset ESPNET_AMD_NO_SA_TREE; fi
      echo -n "[B=$b no_tree=$v] "; bl $b
    done; done 2>&1 | tee "$out/ab_tree.txt"
    unset ESPNET_AMD_NO_SA_TREE
    stats "$out/search640_stats" python "$R/bench.py" --workload beam --batch 64 --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic | cut -c1-200 | head -8 ;;
  parity)   # round 4: the new bf16 parity tests (prints = the measured epsilons), box state, the large encoder's kernel table
    echo "== box state"; BOX_STATE_OUT="$out/box_state" bash tools/box_state.sh 2>&1 | tee "$out/box_state.txt"
    echo "== new parity tests"
    (time timeout 900 python -m pytest -q -s tests/test_gpu_search.py tests/test_gpu_online_search.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py \
       -k "peaked or bf16 or midmargin or structure or lm_scorer_bf16 or restarts" > "$out/pytest_parity_full.txt" 2>&1; grep -E "^\[|passed|failed|Error|assert " "$out/pytest_parity_full.txt" | cut -c1-260 | tail -60) 2>&1 | tee "$out/pytest_parity.txt"
    echo "== large encoder B=64: kernel stats"
    stats "$out/prof_large" python "$R/bench.py" --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 20 --warmup 3
    echo "== default bench"; (time timeout 900 python bench.py 2>"$out/bench_default.err" < /dev/null | tee "$out/bench_default.json" | cut -c1-300) 2>&1 | tail -5 ;;
  r04b)     # stream pool, RecordRing / dynamic dispatch on the GPU (two ranks on one GPU through gloo), parity prints, GEMM stage A/B
    echo "== stream pool + new parity"
    (time timeout 900 python -m pytest -q -s tests/test_gpu_streaming.py tests/test_gpu_search.py tests/test_gpu_online_search.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py \
       -k "pool or peaked or bf16 or midmargin or structure or lm_scorer_bf16 or restarts or batch_call" > "$out/pytest_parity_full.txt" 2>&1; grep -E "^\[|passed|failed|Error|assert " "$out/pytest_parity_full.txt" | cut -c1-260 | tail -60) 2>&1 | tee "$out/pytest_parity.txt"
    echo "== bench quick (RecordRing path)"; timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>"$out/bench_quick.err" < /dev/null | tee "$out/bench_quick.json" | cut -c1-200
    echo "== two ranks on one GPU (gloo): greedy ring + beam dynamic dispatch"
    timeout 300 python bench.py --gpus 2 --dist-debug-one-gpu --quick --no-traffic --no-cpu-baseline --no-roofline --steps 40 --warmup 5 2>"$out/bench_2rank.err" < /dev/null | tee "$out/bench_2rank_greedy.json" | cut -c1-260; tail -3 "$out/bench_2rank.err"
    timeout 400 python bench.py --gpus 2 --dist-debug-one-gpu --workload beam --batch 4 --no-traffic --no-cpu-baseline --steps 3 --warmup 1 2>"$out/bench_2rank_beam.err" < /dev/null | tee "$out/bench_2rank_beam.json" | cut -c1-400; tail -3 "$out/bench_2rank_beam.err"
    echo "== large encoder B=64: GEMM stages A/B"
    for st in 0 2 4; do
      echo -n "stages $st: "; ESPNET_AMD_GEMM_STAGES=$st timeout 200 python bench.py --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null < /dev/null | cut -c100-180
    done ;;
  fold)     # round 4: block<C|D|A> (the C part folded into the launch that consumes it) - unit tests, end-to-end bf16 tests, A/B, table
    echo "== block tests"; timeout 600 python -m pytest tests/test_gpu_block.py -q -x 2>&1 | tail -4 | tee "$out/pytest_block.txt"
    echo "== e2e fused"; timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x -k "bfloat16 or peaked or midmargin or greedy_b32" 2>&1 | tail -4 | tee "$out/pytest_e2e.txt"
    for v in 1 0 1 0; do
      if [ $v = 1 ]; then unset ESPNET_AMD_FOLD; else export ESPNET_AMD_FOLD=1; fi
      echo -n "no_fold=$v: "; timeout 200 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 600 --warmup 30 2>/dev/null < /dev/null | cut -c100-180
    done
    unset ESPNET_AMD_FOLD
    echo "== stamps"; ESPNET_AMD_FOLD=1 EM_BLOCK_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep "block<7>" | tail -2 | tee "$out/block_stamps.txt"
    echo "== kernel stats"; stats "$out/prof_greedy" python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 100 --warmup 10
    echo "== large b64"; timeout 200 python bench.py --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null < /dev/null | cut -c100-180 ;;
  fold-stamps)  # sub-stage stamps of all four waves of one workgroup of block<7> (developer build lib_fine.so)
    ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_fine.so ESPNET_AMD_FOLD=1 EM_BLOCK_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep "block<7>" | tail -8 | tee "$out/block_stamps_fine.txt" ;;
  two-streams)  # probe: the batch as two half batches on two streams
    timeout 300 python tools/two_stream_probe.py 2>&1 | grep "ms per step" | tee "$out/two_stream_probe.txt" ;;
  attn2-large)  # round 4: the large model's attention through attention2 (per-head operands from the projection GEMMs)
    echo "== tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_search.py tests/test_gpu_ebranchformer.py -q -x \
       -k "head_layout or large or gemm or ebf or bf_" 2>&1 | tail -4 | tee "$out/pytest.txt"
    for v in 1 0 1 0; do
      if [ $v = 1 ]; then export ESPNET_AMD_NO_ATTN2_LARGE=1; else unset ESPNET_AMD_NO_ATTN2_LARGE; fi
      echo -n "no_attn2_large=$v: "; timeout 200 python bench.py --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null < /dev/null | cut -c100-180
    done
    unset ESPNET_AMD_NO_ATTN2_LARGE
    echo "== kernel stats"; stats "$out/prof_large" python "$R/bench.py" --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 20 --warmup 3 ;;
  sub2-large)  # round 4: the fused conv1 + conv2 kernel at d = 512 (two launches of 256 output channels)
    echo "== tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "sub12 or large or conv2d" 2>&1 | tail -4 | tee "$out/pytest.txt"
    for v in 1 0 1 0; do
      if [ $v = 1 ]; then export ESPNET_AMD_NO_SUB12=1; else unset ESPNET_AMD_NO_SUB12; fi
      echo -n "no_sub12=$v: "; timeout 200 python bench.py --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null < /dev/null | cut -c100-180
    done
    unset ESPNET_AMD_NO_SUB12
    echo "== kernel stats"; stats "$out/prof_large" python "$R/bench.py" --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 20 --warmup 3 ;;
  ffn-rows)  # round 4: the 512-wide model's feed-forward modules as row-block launches (csrc/ffn_rows.hip)
    echo "== tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_search.py -q -x \
       -k "ffn_rows or large" 2>&1 | tail -6 | tee "$out/pytest.txt"
    for v in 1 0 1 0; do
      if [ $v = 1 ]; then export ESPNET_AMD_NO_FFN_ROWS=1; else unset ESPNET_AMD_NO_FFN_ROWS; fi
      echo -n "no_ffn_rows=$v: "; timeout 200 python bench.py --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null < /dev/null | cut -c100-180
    done
    unset ESPNET_AMD_NO_FFN_ROWS
    echo "== stamps"; EM_FFN_STAMPS=1 timeout 120 python bench.py --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --steps 1 --warmup 1 2>&1 < /dev/null | grep "ffn_rows<" | tail -4 | tee "$out/ffn_stamps.txt"
    echo "== kernel stats"; stats "$out/prof_large" python "$R/bench.py" --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 20 --warmup 3 ;;
  ffn-dbg)  # round 4: where ffn_rows_kernel's time goes: developer builds without MFMAs / weight requests / LDS operand reads, stamps
    bash tools/build_block_variants.sh ffn2 ffn4 ffn6 ffn8 > /dev/null 2>&1
    for v in "" 2 4 6; do
      if [ -z "$v" ]; then unset ESPNET_AMD_LIB; else export ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_ffn$v.so; fi
      echo "-- dbg=${v:-0}"; timeout 120 python tools/ffn_rows_bench.py 2>&1 | grep "ffn_rows mode"
    done 2>&1 | tee "$out/ffn_dbg.txt"
    echo "-- stamps"; ESPNET_AMD_LIB=$R/espnet_amd/lib/dbg/lib_ffn8.so EM_FFN_STAMPS=1 timeout 120 python tools/ffn_rows_bench.py 2>&1 | grep "stamps" | head -4 | tee "$out/ffn_stamps.txt"
    unset ESPNET_AMD_LIB ;;
  stream-fused)  # round 4: contextual-block layer on the row-block kernels (5 launches instead of 13) + E-Branchformer on sub2 / attention2
    echo "== tests"; timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_online_search.py tests/test_gpu_ebranchformer.py tests/test_gpu_block.py -q -x -s 2>&1 | grep -E "^\[stream|passed|failed|Error|assert" | tail -12 | tee "$out/pytest.txt"
    for v in 1 0; do
      if [ $v = 1 ]; then export ESPNET_AMD_STREAM_NO_FUSED=1; else unset ESPNET_AMD_STREAM_NO_FUSED; fi
      echo "stream_no_fused=$v: "; timeout 300 python bench.py --workload stream --no-cpu-baseline 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('  one stream:', d['value'], 'audio-s/s, call latency median', c.get('call_latency_ms_median'), 'p95', c.get('call_latency_ms_p95'))"
    done
    unset ESPNET_AMD_STREAM_NO_FUSED
    echo "== stream leg of the default line (batch32)"; timeout 300 python - <<'PY' 2>/dev/null | tee "$out/stream_batch32.txt"
import json, bench
for v in (True, False):
    import os
    r = bench.run_stream_batch("bfloat16", 32, 3, 1)
    print("batch32:", json.dumps({k: r[k] for k in r if k != "config"})[:300])
    break
PY
    echo "== E-Branchformer"; for v in 1 0; do
      if [ $v = 1 ]; then export ESPNET_AMD_NO_ATTN2_LARGE=1 ESPNET_AMD_NO_SUB12=1; else unset ESPNET_AMD_NO_ATTN2_LARGE ESPNET_AMD_NO_SUB12; fi
      echo -n "old_path=$v: "; timeout 300 python bench.py --model ebf --quick --no-traffic --no-roofline --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null < /dev/null | cut -c100-180
    done
    unset ESPNET_AMD_NO_ATTN2_LARGE ESPNET_AMD_NO_SUB12 ;;
  full|bench)
    if [ "$what" = full ]; then
      echo "== pytest -m gpu"; (time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) 2>&1 | tee "$out/pytest_gpu.txt"
    fi
    echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$out/smoke.txt"
    echo "== default bench"; (time timeout 900 python bench.py 2>"$out/bench_default.err" < /dev/null | tee "$out/bench_default.json" | cut -c1-300) 2>&1 | tail -5
    echo "== rocprofv3 kernel stats, greedy"
    stats "$out/prof_greedy" python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 100 --warmup 10
    if [ "$what" = full ]; then
      echo "== box state (TCC counters)"; BOX_STATE_OUT="$out/box_state" bash tools/box_state.sh 2>&1 | tail -12 | tee "$out/box_state.txt"
      echo "== rocprofv3 kernel stats, large B=64"
      stats "$out/prof_large" python "$R/bench.py" --model large --batch 64 --quick --no-traffic --no-roofline --no-cpu-baseline --in-flight 1 --steps 20 --warmup 3
      echo "== E-Branchformer"; timeout 300 python bench.py --model ebf --quick --no-traffic --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null < /dev/null | tee "$out/bench_ebf.json" | cut -c1-200
      echo "== search kernel stats"; stats "$out/search_stats" python "$R/bench.py" --workload beam --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic
      echo "== search kernel stats, 640 rows (configs[3] per GPU)"; stats "$out/search640_stats" python "$R/bench.py" --workload beam --batch 64 --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic
      echo "== configs[3] per GPU with HBM traffic of the label step"; timeout 600 python bench.py --workload beam --batch 64 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null < /dev/null | tee "$out/bench_beam_b64.json" | cut -c1-300
    fi ;;
  sub2)
    echo "== stamps"
    EM_SUB2_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 3 --warmup 2 2>&1 | grep "sub2 stamps" | head -2 | tee "$out/stamps.txt"
    echo "== tests"; timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "sub12 or conv2d or subsampl" 2>&1 | tail -3
    echo "== kernel stats"; stats "$out/stats" python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 50 --warmup 5
    echo "== bench"; timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>/dev/null < /dev/null | tee "$out/bench_quick.json" | cut -c1-260 ;;
  search-stats)
    stats "$out/search_stats" python "$R/bench.py" --workload beam --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic ;;
  search-pmc)
    cmd="python $R/bench.py --workload beam --steps 1 --warmup 1 --no-cpu-baseline"
    cd /tmp
    echo "== search: kernel stats"
    timeout 400 rocprofv3 --kernel-trace --stats -d "$out/search_stats" -o s --output-format csv -- $cmd > "$out/search_stats.log" 2>&1
    find "$out/search_stats" -name "*_kernel_trace.csv" -delete
    f=$(find "$out/search_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-220
    i=0
    for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      echo "== search: pmc pass $i: $ctrs"
      ESPNET_AMD_SEARCH_GRAPH=0 timeout 600 rocprofv3 --pmc $ctrs --kernel-trace -d "$out/search_pmc$i" -o p --output-format csv -- $cmd > "$out/search_pmc$i.log" 2>&1
      find "$out/search_pmc$i" -name "*_kernel_trace.csv" -delete
      python "$R/tools/pmc_summary.py" "$out/search_pmc$i" --match "" --source "bench.py --workload beam --steps 1 --warmup 1, pass $i" > "$out/search_pmc$i.json" 2>"$out/search_pmc$i.err"
      find "$out/search_pmc$i" -name "*counter_collection.csv" -delete
      head -c 1500 "$out/search_pmc$i.json"
    done
    cd "$R" ;;
  sa-ab)
    for cfg in "1 10" "1 2" "2 2" "2 1" "4 1" "4 2" "8 1"; do
      set -- $cfg
      export ESPNET_AMD_SA_SPLIT=$1 ESPNET_AMD_SA_GROUP=$2
      echo -n "split $1 group $2: "
      timeout 300 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline --no-traffic 2>/dev/null < /dev/null | beam_line
    done ;;
  attn-stamps)
    EM_ATTN2_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 2 --warmup 1 2>&1 < /dev/null | grep "attn2 stamps" | sed -n "13,15p" | tee "$out/attn2_stamps.txt" ;;
  greedy-pmc)
    ctrs="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
    (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace -d "$out/pmc" -o p --output-format csv -- python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 10 --warmup 3 > "$out/pmc.log" 2>&1 < /dev/null)
    find "$out/pmc" -name "*_kernel_trace.csv" -delete 2>/dev/null
    python tools/pmc_summary.py "$out/pmc" --match sub2_kernel relpos_attn2 block_kernel frontend_logmel gemm_kernel --source "bench.py --quick --steps 10 --warmup 3" > "$out/pmc_sq.json" 2>"$out/pmc_sq.err"
    find "$out/pmc" -name "*counter_collection.csv" -delete 2>/dev/null
    head -c 3000 "$out/pmc_sq.json" ;;
  stream-ab)  # round 5: helper workgroups of the row-block launches on a partly filled chip + the contextual-block attention on MFMA:
              # in-call A/B of the switches on the streaming workloads, then the whole suite, the streaming files again with the
              # fused layers taken from ONE block, smoke and the default bench line
    echo "== A/B streaming (one stream: audio-s/s, ms per call; 32 streams: audio-s/s, ms per tick)"
    for cfg in "new" "ESPNET_AMD_BLOCK_NO_HELPERS=1" "ESPNET_AMD_STREAM_MHA_V1=1" "ESPNET_AMD_STREAM_FUSED_MIN=1" "ESPNET_AMD_STREAM_FUSED_MIN=1 ESPNET_AMD_BLOCK_NO_HELPERS=1"; do
      echo -n "$cfg: "
      if [ "$cfg" = new ]; then timeout 200 python tools/stream_ab.py both 2>/dev/null < /dev/null
      else
        case "$cfg" in *FUSED_MIN*) w=one ;; *) w=both ;; esac
        env $cfg timeout 200 python tools/stream_ab.py $w 2>/dev/null < /dev/null
      fi
    done 2>&1 | tee "$out/stream_ab.txt"
    echo "== pytest -m gpu"; (time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) 2>&1 | tee "$out/pytest_gpu.txt"
    echo "== streaming files with the fused layers from one block"
    (ESPNET_AMD_STREAM_FUSED_MIN=1 timeout 300 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_online_search.py -q 2>&1 | tail -6) | tee "$out/pytest_stream_fused_min1.txt"
    echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$out/smoke.txt"
    echo "== default bench"; (time timeout 600 python bench.py 2>"$out/bench_default.err" < /dev/null | tee "$out/bench_default.json" | cut -c1-300) 2>&1 | tail -5
    echo "== kernel table of the 32-stream tick"
    stats "$out/stream32_stats" python "$R/tools/stream_ab.py" batch ;;
  stream-ab2)  # the 128-stream tick (no helper workgroups there: 256 workgroups) with either attention kernel, twice each; the greedy
               # step at batches that leave CUs idle, with and without helper workgroups
    for cfg in "new" "ESPNET_AMD_STREAM_MHA_V1=1" "new" "ESPNET_AMD_STREAM_MHA_V1=1"; do
      echo -n "$cfg: "
      if [ "$cfg" = new ]; then timeout 200 python tools/stream_ab.py batch128 2>/dev/null < /dev/null
      else env $cfg timeout 200 python tools/stream_ab.py batch128 2>/dev/null < /dev/null; fi
    done 2>&1 | tee "$out/stream128_ab.txt"
    for b in 4 8 16; do
      for h in 0 1 0 1; do
        if [ $h = 1 ]; then export ESPNET_AMD_BLOCK_NO_HELPERS=1; else unset ESPNET_AMD_BLOCK_NO_HELPERS; fi
        echo -n "batch $b no_helpers=$h: "
        timeout 200 python bench.py --quick --batch $b --no-traffic --no-roofline --no-cpu-baseline --steps 400 --warmup 30 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms/step')"
      done
    done 2>&1 | tee "$out/helpers_small_batches.txt"
    unset ESPNET_AMD_BLOCK_NO_HELPERS ;;
  stream-ab3)  # CTC.argmax through the arg-max epilogue of the ctc_lo GEMM (no logits), per-tick constants cached
    (timeout 300 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_online_search.py tests/test_gpu_ebranchformer.py -q 2>&1 | tail -4) | tee "$out/pytest_stream.txt"
    (timeout 200 python -m pytest tests/test_gpu_e2e.py -q -k "api or ctc" 2>&1 | tail -3) | tee -a "$out/pytest_stream.txt"
    for cfg in "new" "ESPNET_AMD_CTC_ARGMAX_LOGITS=1" "new" "ESPNET_AMD_CTC_ARGMAX_LOGITS=1"; do
      echo -n "$cfg: "
      if [ "$cfg" = new ]; then timeout 200 python tools/stream_ab.py both 2>/dev/null < /dev/null
      else env $cfg timeout 200 python tools/stream_ab.py both 2>/dev/null < /dev/null; fi
    done 2>&1 | tee "$out/stream_ab3.txt" ;;
  stream-ab4)  # the context hand-over folded into block<A> / block<D> (one block per stream and call)
    (timeout 300 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_block.py tests/test_gpu_online_search.py -q 2>&1 | tail -4) | tee "$out/pytest_stream.txt"
    (ESPNET_AMD_STREAM_FUSED_MIN=1 timeout 300 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_online_search.py -q 2>&1 | tail -4) | tee -a "$out/pytest_stream.txt"
    for cfg in "new" "ESPNET_AMD_STREAM_NO_CTX_FOLD=1" "new" "ESPNET_AMD_STREAM_NO_CTX_FOLD=1"; do
      echo -n "$cfg: "
      if [ "$cfg" = new ]; then timeout 200 python tools/stream_ab.py batch 2>/dev/null < /dev/null
      else env $cfg timeout 200 python tools/stream_ab.py batch 2>/dev/null < /dev/null; fi
    done 2>&1 | tee "$out/stream_ab4.txt" ;;
  final)  # the suite, smoke, the default bench line
    echo "== pytest -m gpu"; (time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) 2>&1 | tee "$out/pytest_gpu.txt"
    echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$out/smoke.txt"
    echo "== default bench"; (time timeout 600 python bench.py 2>"$out/bench_default.err" < /dev/null | tee "$out/bench_default.json" | cut -c1-300) 2>&1 | tail -5 ;;
  *) echo "unknown call: $what"; exit 2 ;;
esac
