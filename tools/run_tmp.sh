mkdir -p gpurun_out/r5
(timeout 300 ./tools/split_mfma_probe) > gpurun_out/r5/split_mfma_probe.txt 2>&1; cat gpurun_out/r5/split_mfma_probe.txt
