#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ summarises (run on the GPU box through gpurun):
#   tools/profile_gpu.sh gpurun_out/profN [sq|all]
# Passes are separate runs, as MI355X_MICROARCH.md prescribes: kernel-trace stats; FETCH_SIZE; WRITE_SIZE; SQ_*.
# Afterwards, here:  python tools/summarize_profiles.py gpurun_out/profN rN
set -e -o pipefail
out=${1:?output directory}
what=${2:-all}
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 5 --no-cpu-baseline"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
if [ "$what" = all ]; then
    timeout -k 10 600 python3 bench.py --steps 20 --warmup 10 > "$out/bench_full.json" 2> "$out/bench_full.err"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- \
        python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline > "$out/bench_stats.log" 2>&1
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o run --output-format csv -- $B > "$out/fetch.log" 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o run --output-format csv -- $B > "$out/write.log" 2>&1
fi
timeout -k 10 300 rocprofv3 --pmc $SQ -d "$out/sq" -o run --output-format csv -- $B > "$out/sq.log" 2>&1
echo "profiles collected in $out"
