#!/usr/bin/env bash
# How the files under profiles/ are produced (run on the GPU box through gpurun, from the repo root):
#
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02_a'
#
# 1. the default bench line                      -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command (kernel durations must agree with the HIP-event
#    means in the bench line)                    -> gpurun_out/<tag>_trace/
# 3. separate --pmc passes for HBM traffic (FETCH_SIZE, WRITE_SIZE may not share a pass), one step of the headline
#    workload at the headline's 100 simulations  -> gpurun_out/<tag>_fetch/, gpurun_out/<tag>_write/
# 4. a --pmc pass with the SQ counters (matrix-pipe busy cycles, wave-cycle breakdown) over stand-alone launches of
#    the two inference kernels at 1 / 2 / 4 workgroups per CU  -> gpurun_out/<tag>_sq/, gpurun_out/<tag>_sq16/
# Afterwards, in the build container:
#   cp gpurun_out/<tag>_trace/*/*_kernel_stats.csv profiles/<tag>_kernel_stats.csv
#   python tools/pmc_summary.py gpurun_out/<tag>_fetch/*/*_counter_collection.csv \
#                               gpurun_out/<tag>_write/*/*_counter_collection.csv profiles/pmc_traffic.json
#   python tools/sq_summary.py gpurun_out/<tag>_sq/*/*_counter_collection.csv --match resnet > profiles/<tag>_mfma_busy.txt
set -euo pipefail
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
python3 "$R/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/${TAG}_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe > "$OUT/${TAG}_write.log" 2>&1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${TAG}_sq" -- python3 "$R/tools/infer_bench.py" --boards 1536 3072 6144 --reps 300 > "$OUT/${TAG}_sq.log" 2>&1
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${TAG}_sq16" -- python3 "$R/tools/infer_bench.py" --boards 1536 3072 6144 --reps 300 --f16 > "$OUT/${TAG}_sq16.log" 2>&1
echo "done: $OUT/${TAG}_*"
