# end-to-end figures of two library builds, interleaved: the sustained pipeline (survey3_65536 lists) and the corpus by title
for i in 1 2; do for v in "$@"; do
  lib=$PWD/dcsexplorer_amd/libdcs_hip_$v.so; [ $v = ship ] && lib=$PWD/dcsexplorer_amd/libdcs_hip.so
  DCS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-device-path --no-second-workload --no-class-surface --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['end_to_end']
print('$v sustained %.4f ms/list  dev-index %.4f  cold %.3f' % (e['sustained']['ms_per_list'], e['sustained_device_index']['ms_per_list'], e['cold']['ms_per_list']))"
  DCS_HIP_LIB=$lib python bench.py --workload corpus --corpus-streams 600 --no-cpu-baseline --no-device-path --no-second-workload --no-class-surface --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['end_to_end']['corpus_by_title']
print('$v corpus_by_title first %.4f s second %.4f s' % (c['first_pass']['seconds'], c['second_pass']['seconds']))"
done; done
