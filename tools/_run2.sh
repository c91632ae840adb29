export TMPDIR=/tmp; o=gpurun_out
rm -rf $o/r06_tl; mkdir -p $o/r06_tl
rocprofv3 --kernel-trace --output-format csv -d $o/r06_tl -- python3 tools/update_rate.py --configs small --modes fused --reps 2 > /dev/null 2>&1
python3 tools/timeline.py $o/r06_tl --dump 40 > $o/r06_timeline_update.txt 2>&1
rm -rf $o/r06_tl
cat $o/r06_timeline_update.txt
