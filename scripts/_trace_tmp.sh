mkdir -p gpurun_out/pp2
PROBE_CORPUS=docs_1m python scripts/c5_probe.py uni32k 64 "" > gpurun_out/pp2/docs1m.txt 2>&1
python -m pytest tests/test_documents.py tests/test_gpu_parity.py -m gpu -x -q -k "document or long" > gpurun_out/pp2/pytest.txt 2>&1; tail -2 gpurun_out/pp2/pytest.txt
cat gpurun_out/pp2/docs1m.txt
