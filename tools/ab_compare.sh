#!/bin/bash
# Same-box A/B of library builds: tools/ab_compare.sh ab/libldpc_A.so ab/libldpc_B.so ...   (run through gpurun)
# Each variant is copied over libldpc_amd/libldpc.so in the box's scratch copy of the repo and measured twice,
# interleaved, with tools/standalone_probe.py (decode kernel alone and pipelined).
set -e
cp libldpc_amd/libldpc.so /tmp/libldpc_orig.so
for round in 1 2; do
  for v in "$@"; do
    cp "$v" libldpc_amd/libldpc.so
    echo "== $v (round $round)"
    timeout -k 10 300 python3 tools/standalone_probe.py 2>&1 | grep -v amdgpu
  done
done
cp /tmp/libldpc_orig.so libldpc_amd/libldpc.so
