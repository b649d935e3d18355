#!/bin/bash
# Collect the rocprofv3 evidence for the solve path on the GPU box.
#   scripts/collect_profiles.sh <tag> [bench.py arguments]     -> gpurun_out/prof_<tag>/...
# Each profiler pass is its own run (counters never share a run with traces; FETCH_SIZE and WRITE_SIZE do
# not fit one pass).  The calibration passes run one k_solve launch whose byte count is known exactly.
set -e -o pipefail
TAG=${1:-cur}
shift || true
ARGS="$@"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o stats -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > "$OUT/bench_stats.log" 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > "$OUT/bench_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o write -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > "$OUT/bench_write.log" 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/sq" -o sq -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > "$OUT/bench_sq.log" 2>&1
for CB in 16384 131072; do
    timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/cal_fetch_$CB" -o cal -- \
        python3 scripts/fetch_calibration.py $CB > "$OUT/cal_fetch_$CB.json" 2> "$OUT/cal_fetch_$CB.log"
    timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/cal_write_$CB" -o cal -- \
        python3 scripts/fetch_calibration.py $CB > "$OUT/cal_write_$CB.json" 2> "$OUT/cal_write_$CB.log"
done
find "$OUT" -name '*.csv' | sort
