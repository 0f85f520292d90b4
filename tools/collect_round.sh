#!/bin/bash
# Everything profiles/ holds for a round, on the GPU box (through gpurun; the whole of it exceeds one call's 20 minutes):
#   tools/collect_round.sh gpurun_out/rN a      bench.py lines of every configuration (roofline + cpu_baseline) -> configs.jsonl
#   tools/collect_round.sh gpurun_out/rN b      rocprofv3 --kernel-trace --stats of the default bench.py command -> stats/,
#                                               shard cost at world sizes 1..8 (tools/shard_probe.py), the non-parity modes'
#                                               error-rate report (tools/fast_mode_report.py), per-configuration counter passes
#                                               (tools/profile_config.sh: SQ sets, instruction mix, FETCH_SIZE, WRITE_SIZE, stats)
# then here:  python tools/summarize_profiles.py gpurun_out/rN rN
set -o pipefail
out=${1:?output directory}
part=${2:-all}
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$part" = a ] || [ "$part" = all ]; then
    : > "$out/configs.jsonl"
    for c in 2 1 3 4 5 5bec 2n 4n 2f 2l 2h; do
        timeout -k 10 300 python3 bench.py --config $c --steps 40 --warmup 10 2> "$out/bench_cfg$c.err" | tail -1 >> "$out/configs.jsonl" || echo "bench config $c failed"
        echo "config $c done"
    done
fi
if [ "$part" = b ] || [ "$part" = all ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- \
        python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-pmc --no-other-configs > "$out/bench_stats.log" 2>&1 || echo "stats pass failed"
    find "$out/stats" -name "*kernel_trace.csv" -delete
    timeout -k 10 300 python3 tools/shard_probe.py --config 2 --worlds 1,2,4,8 --out "$out/shard_cost_cfg2.jsonl" > "$out/shard2.log" 2>&1 || echo "shard probe (config 2) failed"
    timeout -k 10 300 python3 tools/shard_probe.py --config 4 --worlds 1,2,4,8 --out "$out/shard_cost_cfg4.jsonl" > "$out/shard4.log" 2>&1 || echo "shard probe (config 4) failed"
    timeout -k 10 300 python3 tools/shard_probe.py --config 2 --gen --worlds 1,2,4,8 --steps 30 --out "$out/shard_cost_cfg2_gen.jsonl" > "$out/shard2g.log" 2>&1 || echo "shard probe (config 2, -G) failed"
    echo "probes done"
    timeout -k 10 500 python3 tools/fast_mode_report.py > "$out/fast_mode_report.jsonl" 2> "$out/fast_mode_report.err" || echo "fast mode report failed"
    echo "fast mode report done"
    for c in 2 3 4 5 5bec 2n 4n; do
        tools/profile_config.sh "$out" $c 3 || echo "profile of config $c failed"
        find "$out/cfg$c" -name "*kernel_trace.csv" -delete
    done
fi
echo "round collected in $out ($part)"
