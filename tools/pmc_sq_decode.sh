#!/bin/bash
# SQ counter passes over the DECODE kernels at the bench's batch size (tools/kbench.py --frames 15504 --decode)
# usage (on the GPU box): bash tools/pmc_sq_decode.sh <outdir>; then python3 tools/pmc_sq_summary.py <outdir> <out.json>
set -e
out=${1:-gpurun_out/pmc_sq_dec}
mkdir -p "$out"
export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$out/$name" -o p --output-format csv -- python3 tools/kbench.py --frames 15504 --reps 1 --no-timing --decode > "$out/$name.log" 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAVES
