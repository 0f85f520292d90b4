#!/bin/bash
# Same-box A/B over environment settings as well as library builds (through gpurun):
#   tools/ab_env.sh <config> <reps> "<ENV=VAL ...|->" lib [ "<env>" lib ... ]
set -o pipefail
c=${1:?config}; reps=${2:?repetitions}; shift 2
for rep in $(seq 1 $reps); do
  args=("$@")
  for ((i = 0; i < ${#args[@]}; i += 2)); do
    e=${args[i]}; v=${args[i+1]}
    [ "$e" = "-" ] && e=""
    echo -n "$v [$e] cfg$c rep$rep: "
    env $e LDPC_AMD_LIB="$PWD/$v" timeout -k 10 300 python3 bench.py --config $c --steps 100 --warmup 12 --no-pmc --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
s = j.get('step_ms') or {}
print('ms_per_step %.3f kernel_ms %.3f rng_ms %.3f median %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['rng_ms_avg'], s.get('median_max_over_ranks', 0)))"
  done
done
