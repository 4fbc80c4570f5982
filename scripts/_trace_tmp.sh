mkdir -p gpurun_out/uw7
PROBE_CORPUS=docs_1m python scripts/c5_probe.py uni32k 64 "" > gpurun_out/uw7/docs1m.txt 2>&1
PROBE_CORPUS=docs_16k python scripts/c5_probe.py uni32k 8192 "" > gpurun_out/uw7/docs16k.txt 2>&1
python -m pytest tests -m gpu -x -q -k "long or doc or uniwave or wave or full_size or parity" > gpurun_out/uw7/pytest.txt 2>&1
tail -3 gpurun_out/uw7/pytest.txt
cat gpurun_out/uw7/docs16k.txt gpurun_out/uw7/docs1m.txt
