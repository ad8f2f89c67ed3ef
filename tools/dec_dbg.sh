#!/bin/bash
for d in 0 201; do
  LINNE_AMD_DBG_MAXTR=$d python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $d', {k[:24]:v for k,v in d['kernel_ms_per_step'].items() if 'synth' in k or 'ms_to' in k or 'deemph' in k})"
done
