mkdir -p gpurun_out/r5
(timeout 900 python -m pytest tests/test_c2_parity.py -q -x -s -k chain 2>&1 | grep -v "^$" | tail -25) > gpurun_out/r5/chain_test.txt; cat gpurun_out/r5/chain_test.txt
