#!/bin/bash
# several builds of the library on ONE box, one after the other, `rounds` times: exclusive per-kernel spans of bench.py's one-stream step
# usage (on the GPU box): bash tools/ab3.sh rounds libA.so libB.so libC.so ...      [AB3_KEY=substring of the kernel names to print]
rounds=$1; shift
for r in $(seq 1 $rounds); do for lib in "$@"; do
  LINNE_AMD_LIB=$PWD/$lib python3 bench.py --steps 4 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline'].get('exclusive_kernel_ms', {}); key=os.environ.get('AB3_KEY','hist'); print('$lib', 'step', round(d['ms_per_step'],2), 'one stream', d['roofline'].get('one_stream_step_ms'), {k[:24]:v for k,v in e.items() if key in k}, 'decode', round(d['decode_ms_per_step'],3), d['decode_bit_exact'], flush=True)"
done; done
