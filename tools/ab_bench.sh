#!/bin/bash
# Same-box A/B of library builds on whole bench.py steps (side stream included), through gpurun:
#   tools/ab_bench.sh "<configs>" <reps> ab/libldpc_A.so ab/libldpc_B.so ...
# The variant is selected with LDPC_AMD_LIB (libldpc_amd/binding.py); variants interleaved, 100 timed steps each.
set -o pipefail
configs=${1:?configs}; reps=${2:?repetitions}; shift 2
for rep in $(seq 1 $reps); do
  for c in $configs; do
    for v in "$@"; do
      echo -n "$v cfg$c rep$rep: "
      LDPC_AMD_LIB="$PWD/$v" timeout -k 10 300 python3 bench.py --config $c --steps 100 --warmup 12 --no-pmc --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
s = j.get('step_ms') or {}
print('ms_per_step %.3f kernel_ms %.3f rng_ms %.3f median %.3f max %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['rng_ms_avg'], s.get('median_max_over_ranks', 0), s.get('max', 0)))"
    done
  done
done
