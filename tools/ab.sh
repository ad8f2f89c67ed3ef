#!/bin/bash
# A/B of two builds of the library on ONE box (boxes of the pool differ by 1-2 %): alternates LINNE_AMD_LIB between the files given
# (per-kernel figures: the exclusive spans of bench.py's extra one-stream step)
# usage (on the GPU box): bash tools/ab.sh linne_amd/ab/libA.so linne_amd/ab/libB.so [rounds] -- prints ms per step and the per-kernel times
rounds=${3:-2}
for r in $(seq 1 $rounds); do for lib in "$1" "$2"; do
  LINNE_AMD_LIB=$PWD/$lib python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],2), 'one stream', d['roofline'].get('one_stream_step_ms'), {k[:22]:v for k,v in d['roofline'].get('exclusive_kernel_ms', {}).items() if v > 1.0}, flush=True)"
done; done
