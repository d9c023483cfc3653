for i in 1 2 3; do for full in 0 1; do
  DCS_EXP_FULL_IMAGE=$full python bench.py --steps 200 --no-cpu-baseline --no-end-to-end --no-device-path --no-second-workload 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('full_image=$full warm %.2f us cold %.2f us pkg %s bit_exact %s cold_exact %s' % (d['roofline']['kernel_avg_ms']*1e3, d['roofline_cold']['kernel_avg_ms']*1e3, d['roofline_cold']['resident_bytes_rotated'], d['bit_exact'], d['roofline_cold']['bit_exact']))"
done; done
