#!/bin/bash
# Same-box A/B of library builds (run through gpurun):  tools/ab.sh "<configs>" ab/libldpc_A.so ab/libldpc_B.so ...
# The variant is selected with LDPC_AMD_LIB (libldpc_amd/binding.py); the product library is never overwritten.
# Each variant is measured twice, interleaved; prints kernel ms per batch from the library's HIP events.
set -e -o pipefail
configs=${1:?configs, e.g. "2 4"}
shift
for round in 1 2; do
  for v in "$@"; do
    for c in $configs; do
      echo -n "$v cfg$c (round $round): "
      LDPC_AMD_LIB="$PWD/$v" timeout -k 10 300 python3 tools/pmc_probe.py --config "$c" --steps 6 --warmup 3 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.3f  eu/s(kernel) %.3e' % (j['kernel_ms'], j['edge_updates'] / j['steps'] / j['kernel_ms'] * 1e3))"
    done
  done
done
