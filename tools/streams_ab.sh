for r in 1 2; do for s in 2 3 4; do
  LINNE_AMD_STREAMS=$s python3 bench.py --steps 5 --no-end-to-end --no-transports --no-cpu-baseline --no-block-at-a-time --no-sample-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s', round(d['ms_per_step'],2), round(d['value']), d['decode_bit_exact'])"
done; done
