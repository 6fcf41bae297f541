#!/usr/bin/env bash
# How the files under profiles/ are produced (run on the GPU box through gpurun, from the repo root):
#
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03_a'
#
# 1. the default bench line                      -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same workload (kernel durations must agree with the HIP-event means in
#    the bench line): once without the evaluation-cache runs (the headline's and the f16x2 line's kernels), once
#    with only the evaluation-cache run (same kernel names, other launch shapes)      -> <tag>_kernel_stats.csv, <tag>_cache_kernel_stats.csv
# 3. separate --pmc passes for HBM traffic (FETCH_SIZE, WRITE_SIZE may not share a pass), one step of the headline
#    workload at the headline's 100 simulations  -> gpurun_out/<tag>_fetch/, gpurun_out/<tag>_write/
# 4. a --pmc pass with the SQ counters (matrix-pipe busy cycles, wave-cycle breakdown) over stand-alone launches of
#    the two inference kernels at 1 / 2 / 4 workgroups per CU  -> gpurun_out/<tag>_sq/, gpurun_out/<tag>_sq16/
# The summaries (tools/pmc_summary.py, tools/sq_summary.py, the *_kernel_stats.csv of each trace) are made on the box and
# the raw per-dispatch files deleted (gpurun brings back at most 64 MiB).  Afterwards, in the build container:
#   cp gpurun_out/<tag>_{bench.json,kernel_stats.csv,cache_*,pmc_traffic.json,mfma_busy.txt,config3_*,config4_*,config5_*,*games_bench.json,refdefault_bench.json,config1_bench.json,onegame_bench.json} profiles/
#   cp gpurun_out/<tag>_pmc_traffic.json profiles/pmc_traffic.json        (what bench.py's roofline.traffic reads)
set -euo pipefail
TAG=${1:-prof}
PART=${2:-AB}          # A = steps 1-4 (headline), B = step 5 (other configurations): one gpurun call each (20-minute limit)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
keep_stats() { cp "$OUT/$1"/*/*_kernel_stats.csv "$OUT/$2" && rm -rf "$OUT/$1"; }      # the per-dispatch trace is tens of MB
if [[ $PART == *A* ]]; then
python3 "$R/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$R/bench.py" --no-cpu-baseline --no-eval-cache > "$OUT/${TAG}_trace.log" 2>&1
keep_stats "${TAG}_trace" "${TAG}_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_cache_trace" -- python3 "$R/bench.py" --no-cpu-baseline --no-clock-probe --single --eval-cache-headline > "$OUT/${TAG}_cache_bench.json" 2> "$OUT/${TAG}_cache_trace.log"
keep_stats "${TAG}_cache_trace" "${TAG}_cache_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-eval-cache > "$OUT/${TAG}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-eval-cache > "$OUT/${TAG}_write.log" 2>&1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
python3 "$R/tools/pmc_summary.py" "$OUT/${TAG}_fetch"/*/*_counter_collection.csv "$OUT/${TAG}_write"/*/*_counter_collection.csv "$OUT/${TAG}_pmc_traffic.json" > "$OUT/${TAG}_pmc_summary.log" 2>&1
rm -rf "$OUT/${TAG}_fetch" "$OUT/${TAG}_write"
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${TAG}_sq" -- python3 "$R/tools/infer_bench.py" --boards 1536 3072 6144 --reps 300 > "$OUT/${TAG}_sq.log" 2>&1
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${TAG}_sq16" -- python3 "$R/tools/infer_bench.py" --boards 1536 3072 6144 --reps 300 --f16 > "$OUT/${TAG}_sq16.log" 2>&1
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${TAG}_sqg9" -- python3 "$R/tools/infer_bench.py" --game gomoku9 --blocks 10 --channels 256 --boards 2048 --reps 6 > "$OUT/${TAG}_sqg9.log" 2>&1
{ echo "== exact fp32 kernel (1536 / 3072 / 6144 boards = 1 / 2 / 4 workgroups per CU)"; python3 "$R/tools/sq_summary.py" "$OUT/${TAG}_sq"/*/*_counter_collection.csv --match resnet;
  echo "== split-fp16 kernel"; python3 "$R/tools/sq_summary.py" "$OUT/${TAG}_sq16"/*/*_counter_collection.csv --match resnet;
  echo "== Gomoku 9x9, 10 x 256, 2048 boards"; python3 "$R/tools/sq_summary.py" "$OUT/${TAG}_sqg9"/*/*_counter_collection.csv --match resnet; } > "$OUT/${TAG}_mfma_busy.txt" 2>&1
rm -rf "$OUT/${TAG}_sq" "$OUT/${TAG}_sq16" "$OUT/${TAG}_sqg9"
fi
if [[ $PART == *B* ]]; then
cd /tmp && export TMPDIR=/tmp
# 5. the other single-GPU configurations: config 3 (8192 games x 400 simulations), config 4's network (8 x 128, one GPU's
#    4096-game shard) and config 5 (Gomoku 9x9, 10 x 256, 200 simulations, 512 games), each as a bench line and a kernel
#    trace of one step; then the small runs: what the reference's own entry point calls (alphazero_less_simple/main.py:18-28:
#    ResNet(9, 128), 100 simulations, 100 games) and BASELINE config 1's shape (one game, 25 simulations)
python3 "$R/bench.py" --slots 8192 --sims 400 --steps 1 --warmup 1 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_config3_bench.json" 2> "$OUT/${TAG}_config3_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_config3_trace" -- python3 "$R/bench.py" --slots 8192 --sims 400 --single --no-eval-cache --no-cpu-baseline --no-clock-probe --steps 1 --warmup 0 > "$OUT/${TAG}_config3_trace.log" 2>&1
keep_stats "${TAG}_config3_trace" "${TAG}_config3_kernel_stats.csv"
python3 "$R/bench.py" --slots 100 --blocks 9 --channels 128 --steps 3 --warmup 1 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_refdefault_bench.json" 2> "$OUT/${TAG}_refdefault_bench.err"
python3 "$R/bench.py" --slots 100 --steps 3 --warmup 1 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_100games_bench.json" 2> "$OUT/${TAG}_100games_bench.err"
python3 "$R/bench.py" --slots 1 --sims 25 --steps 20 --warmup 2 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_config1_bench.json" 2> "$OUT/${TAG}_config1_bench.err"
python3 "$R/bench.py" --slots 1 --steps 5 --warmup 1 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_onegame_bench.json" 2> "$OUT/${TAG}_onegame_bench.err"
python3 "$R/bench.py" --blocks 8 --channels 128 --no-cpu-baseline > "$OUT/${TAG}_config4_bench.json" 2> "$OUT/${TAG}_config4_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_config4_trace" -- python3 "$R/bench.py" --blocks 8 --channels 128 --single --no-eval-cache --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/${TAG}_config4_trace.log" 2>&1
keep_stats "${TAG}_config4_trace" "${TAG}_config4_kernel_stats.csv"
python3 "$R/bench.py" --game gomoku9 --blocks 10 --channels 256 --sims 200 --slots 512 --no-cpu-baseline --steps 1 --warmup 1 > "$OUT/${TAG}_config5_bench.json" 2> "$OUT/${TAG}_config5_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_config5_trace" -- python3 "$R/bench.py" --game gomoku9 --blocks 10 --channels 256 --sims 200 --slots 512 --single --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/${TAG}_config5_trace.log" 2>&1
keep_stats "${TAG}_config5_trace" "${TAG}_config5_kernel_stats.csv"
fi
echo "done: $OUT/${TAG}_*"
