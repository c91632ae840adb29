export TMPDIR=/tmp; o=gpurun_out
bash tools/make_profiles.sh r06 $1 lanes lengths check > $o/r06_make_profiles3.log 2>&1
tail -4 $o/r06_lanes_two_streams.txt | cut -c1-300
python3 -c "
import json
for n in ('lengthslognormal','uniform','driver','200'):
    j=json.load(open('$o/r06_bench_%s.json'%n)); print(n, j['ms_per_step'], j['value'], j.get('lane_state'), j.get('lane_calibration'), (j.get('value_end_to_end') or {}).get('value'), ((j.get('value_end_to_end') or {}).get('one_call') or {}).get('value'))"
cat $o/r06_final_check.txt | tail -6 | cut -c1-400
