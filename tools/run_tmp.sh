O=gpurun_out/r5; mkdir -p $O
(timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15) > $O/gputest_a.txt
cat $O/gputest_a.txt
