#!/bin/bash
# rocprofv3 evidence for one BASELINE configuration (run on the GPU box through gpurun):
#   tools/profile_config.sh gpurun_out/profN <config> [steps]
# Separate passes, as MI355X_MICROARCH.md prescribes: kernel-trace stats; SQ set 1; SQ set 2; the instruction mix; FETCH_SIZE;
# WRITE_SIZE.
# The workload is tools/pmc_probe.py (ctypes only, no torch): the profiler sees the library's kernels and nothing else.
# Afterwards, here:  python tools/summarize_profiles.py gpurun_out/profN rN
set -e -o pipefail
out=${1:?output directory}
cfg=${2:?configuration}
steps=${3:-3}
mkdir -p "$out"
export TMPDIR=/tmp
P="python3 tools/pmc_probe.py --config $cfg --steps $steps --warmup 2"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
MIX="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVES"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/cfg$cfg/stats" -o run --output-format csv -- $P > "$out/cfg$cfg.stats.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc $SQ1 -d "$out/cfg$cfg/sq" -o run --output-format csv -- $P > "$out/cfg$cfg.sq1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc $SQ2 -d "$out/cfg$cfg/sq2" -o run --output-format csv -- $P > "$out/cfg$cfg.sq2.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc $MIX -d "$out/cfg$cfg/mix" -o run --output-format csv -- $P > "$out/cfg$cfg.mix.log" 2>&1 || echo "mix pass of config $cfg failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$out/cfg$cfg/fetch" -o run --output-format csv -- $P > "$out/cfg$cfg.fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$out/cfg$cfg/write" -o run --output-format csv -- $P > "$out/cfg$cfg.write.log" 2>&1
echo "profiles of config $cfg collected in $out/cfg$cfg"
