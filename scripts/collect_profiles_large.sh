#!/bin/bash
# rocprofv3 evidence for the large-state path (bench.py --workload large): kernel stats, one SQ counter pass,
# and the two HBM traffic passes.
#   scripts/collect_profiles_large.sh <tag>      -> gpurun_out/prof_<tag>/...
set -e -o pipefail
TAG=${1:-large}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o stats -- \
    python3 bench.py --workload large --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/sq" -o sq -- \
    python3 bench.py --workload large --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_sq.log" 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- \
    python3 bench.py --workload large --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o write -- \
    python3 bench.py --workload large --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_write.log" 2>&1
find "$OUT" -name '*.csv' | sort
