#!/bin/bash
# decode step under several builds of the library on one box (timing experiments: results of libexp* are wrong)
for lib in "$@"; do
  LINNE_AMD_LIB=$PWD/$lib python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'decode ms', round(d['decode_ms_per_step'],2), d.get('decode_bit_exact'), {k[:20]:v for k,v in d['kernel_ms_per_step'].items() if 'synth' in k or 'ms_to' in k or 'deemph' in k}, flush=True)"
done
