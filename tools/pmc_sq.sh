#!/bin/bash
# SQ counter passes over tools/kbench.py (4096 stereo frames, one encode call): per-kernel issue/LDS/wait picture.
# usage (on the GPU box): bash tools/pmc_sq.sh <outdir>      -- each pass is its own rocprofv3 run (counters only with --kernel-trace)
set -e
out=${1:-gpurun_out/pmc_sq}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$out/$name" -o p --output-format csv -- python3 tools/kbench.py --frames 4096 --reps 1 --no-timing > "$out/$name.log" 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAVES
