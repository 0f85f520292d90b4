#!/bin/bash
# Same-box A/B of library builds on the headline step time: tools/ab_bench.sh ab/libldpc_X.so ...  (through gpurun)
set -e
cp libldpc_amd/libldpc.so /tmp/libldpc_orig.so
for round in 1 2; do
  for v in "$@"; do
    cp "$v" libldpc_amd/libldpc.so
    echo -n "$v (round $round): "
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 | grep -o "ms_per_step[^,]*\|kernel_ms_avg[^,]*\|rng_ms_avg[^,]*" | tr '\n' ' '
    echo
  done
done
cp /tmp/libldpc_orig.so libldpc_amd/libldpc.so
