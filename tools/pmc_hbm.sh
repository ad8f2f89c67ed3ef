#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (MI355X_MICROARCH.md) over tools/kbench.py
# (4096 stereo frames, one encode + one decode call), then tools/pmc_traffic.py -> profiles/pmc_latest.json
# usage (on the GPU box): bash tools/pmc_hbm.sh <outdir>
set -e
out=${1:-gpurun_out/pmc_hbm}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (the 4096 frames are 32 768 jobs: LINNE_AMD_LAST_LAYER=2 gives them the kernel set of the bench's 62 016-job halves -- k_last_layer from 24 576 jobs on instead of 49 152)
export LINNE_AMD_LAST_LAYER=2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/f" -o p --output-format csv -- python3 tools/kbench.py --frames 4096 --reps 1 --no-timing --decode > "$out/f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/w" -o p --output-format csv -- python3 tools/kbench.py --frames 4096 --reps 1 --no-timing --decode > "$out/w.log" 2>&1
python3 tools/pmc_traffic.py "$out/f/p_counter_collection.csv" "$out/w/p_counter_collection.csv" 4096 2 "${2:-}"
cp profiles/pmc_latest.json "$out/pmc_latest.json"
