#!/bin/bash
# One GPU call that regenerates the round's measured artefacts under gpurun_out/<tag>/ (copied into profiles/ afterwards):
#   bench.json (default driver-style run), kernel_stats.csv (rocprofv3 --kernel-trace --stats of the timed region only),
#   pmc_latest.json (HBM FETCH / WRITE passes), sq_summary.json (SQ counter passes), block_timeline.txt (one EncodeBlock / DecodeBlock),
#   stress8ch.json (BASELINE configs[4]), mg2.json (two ranks on one GPU, gloo transfers: a rehearsal of the N > 1 path)
#   kernel_stats_one_stream.csv (the same with LINNE_AMD_STREAMS=1: exclusive per-kernel durations -- what bench.py's roofline is priced on),
#   sq_decode.json (SQ counters of the decode kernels at the bench's batch size)
# usage (on the GPU box): bash tools/round_profiles.sh <tag> <commit id of the tree>   e.g.  gpurun -- 'bash tools/round_profiles.sh r04 '"$(git rev-parse --short HEAD)"
tag=${1:-r04}; note=${2:-}
out=gpurun_out/$tag
mkdir -p "$out"; export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc $?"
rocprofv3 --kernel-trace --stats -d "$out/ks" -o p --output-format csv -- python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity > "$out/bench_profiled_run.json" 2> "$out/ks.err"
cp "$out/ks/p_kernel_stats.csv" "$out/kernel_stats.csv" 2>/dev/null; rm -rf "$out/ks"
LINNE_AMD_STREAMS=1 rocprofv3 --kernel-trace --stats -d "$out/ks1" -o p --output-format csv -- python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity > "$out/bench_profiled_run_one_stream.json" 2> "$out/ks1.err"
cp "$out/ks1/p_kernel_stats.csv" "$out/kernel_stats_one_stream.csv" 2>/dev/null; rm -rf "$out/ks1"
bash tools/pmc_hbm.sh "$out/pmc" "$note" > "$out/pmc.log" 2>&1; rm -rf "$out/pmc/f" "$out/pmc/w"
bash tools/pmc_sq.sh "$out/sq" > "$out/sq.log" 2>&1; python3 tools/pmc_sq_summary.py "$out/sq" "$out/sq_summary.json" > "$out/sq_summary.txt" 2>&1; rm -rf "$out/sq/a" "$out/sq/b"
bash tools/pmc_sq_decode.sh "$out/sqd" > "$out/sqd.log" 2>&1; python3 tools/pmc_sq_summary.py "$out/sqd" "$out/sq_decode.json" > "$out/sq_decode.txt" 2>&1; rm -rf "$out/sqd/a" "$out/sqd/b"
rocprofv3 --kernel-trace -d "$out/bt" -o p --output-format csv -- python3 tools/blockrate.py 30 > "$out/bt.log" 2>&1; python3 tools/block_timeline.py "$out/bt" > "$out/block_timeline.txt" 2>&1; rm -rf "$out/bt"
python3 bench.py --channels 8 --bits 24 --rate 96000 --minutes 10 --no-transports > "$out/stress8ch.json" 2> "$out/stress8ch.err"
BENCH_BACKEND=gloo python3 bench.py --gpus 2 --minutes 10 --steps 3 > "$out/mg2.json" 2> "$out/mg2.err"; echo "mg2 rc $?"
ls -la "$out"
