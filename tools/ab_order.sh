for v in "" "DCS_EXP_CHAIN_ORDER=1" "DCS_BATCH_XCD_RANGES=1"; do for i in 1 2 3; do env $v python bench.py --no-class-surface --no-end-to-end --no-cpu-baseline --no-device-path --no-second-workload --rotate 0 --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value', round(d['value']/1e11,3), 'kernel_us', round(d['roofline']['kernel_avg_ms']*1e3,2), d['bit_exact'])
"; done; done
