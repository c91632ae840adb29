export TMPDIR=/tmp; o=gpurun_out
timeout 3000 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $o/r06_t4.txt
for sd in 1 2 3; do timeout 900 python tests/fuzz_lifecycle.py --steps 300 --seed $sd 2>&1 | tail -1; done > $o/r06_fuzz_lifecycle.txt
timeout 600 python tests/fuzz_deferred.py --lanes 2>&1 | tail -2 >> $o/r06_fuzz_lifecycle.txt
cat $o/r06_t4.txt $o/r06_fuzz_lifecycle.txt | cut -c1-200
