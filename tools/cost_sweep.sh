for c in 2,2 1,4 1,8 1,16 2,16 1,32; do
  echo -n "VN_COST=$c: "
  LDPC_AMD_VN_COST=$c timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 | grep -o "ms_per_step[^,]*\|kernel_ms_avg[^,]*" | tr '\n' ' '
  echo
done
echo -n "VN_COST=2,2 again: "
LDPC_AMD_VN_COST=2,2 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 | grep -o "ms_per_step[^,]*\|kernel_ms_avg[^,]*" | tr '\n' ' '
