#!/bin/bash
# decode step of bench.py under each form of the synthesis (one box): ms per kernel kind
# (default: k_synth_l0_de closes the cascade; unfused: LINNE_AMD_DECODE_FUSED=0 -- k_synth_rows8 + k_deemph_lr as launches of their own)
for k in default unfused default unfused lanes; do
  unset LINNE_AMD_DECODE_KERNEL LINNE_AMD_DECODE_FUSED
  if [ $k = unfused ]; then export LINNE_AMD_DECODE_FUSED=0; elif [ $k != default ]; then export LINNE_AMD_DECODE_KERNEL=$k; fi
  python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'decode fps', d.get('decode_frames_per_s'), 'bit_exact', d.get('decode_bit_exact'), 'ms', round(d.get('decode_ms_per_step'),2), {k[:24]:v for k,v in d['kernel_ms_per_step'].items() if 'synth' in k or 'ms_to' in k or 'deemph' in k}, flush=True)"
done
