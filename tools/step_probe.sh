#!/bin/bash
# whole-step A/B (noise stream running beside the decode kernel, as in bench.py): tools/step_probe.sh "<configs>" lib.so ...
configs=${1:?configs}; shift
for round in 1 2; do for v in "$@"; do for c in $configs; do
  echo -n "$v cfg$c (round $round): "
  LDPC_AMD_LIB="$PWD/$v" timeout -k 10 200 python3 bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('ms/step %.3f  kernel %.3f  value %.4g'%(j['ms_per_step'], j['roofline'].get('kernel_ms_avg') or 0, j['value']))"
done; done; done
