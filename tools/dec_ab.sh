#!/bin/bash
# decode step of bench.py under each form of the synthesis (one box): ms per kernel kind
for k in default lanes rows; do
  if [ $k = default ]; then unset LINNE_AMD_DECODE_KERNEL; else export LINNE_AMD_DECODE_KERNEL=$k; fi
  python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'decode fps', d.get('decode_frames_per_s'), 'bit_exact', d.get('decode_bit_exact'), 'ms', round(d.get('decode_ms_per_step'),2), {k[:24]:v for k,v in d['kernel_ms_per_step'].items() if 'synth' in k or 'ms_to' in k or 'deemph' in k})"
done
