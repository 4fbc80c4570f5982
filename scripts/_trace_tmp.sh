mkdir -p gpurun_out/uw15
PROBE_CORPUS=docs_1m python scripts/c5_probe.py uni32k 64 "" > gpurun_out/uw15/docs1m.txt 2>&1
PROBE_CORPUS=docs_16k python scripts/c5_probe.py uni32k 8192 "" > gpurun_out/uw15/docs16k.txt 2>&1
python scripts/c5_probe.py c5_250k 1000000 "" > gpurun_out/uw15/c5.txt 2>&1
cat gpurun_out/uw15/docs16k.txt gpurun_out/uw15/docs1m.txt gpurun_out/uw15/c5.txt
