mkdir -p gpurun_out/pp
PROBE_CORPUS=docs_1m python scripts/c5_probe.py uni32k 64 "" "SPMX_UW_PIPE=0" > gpurun_out/pp/docs1m.txt 2>&1
PROBE_CORPUS=docs_16k python scripts/c5_probe.py uni32k 8192 "" "SPMX_UW_PIPE=2" > gpurun_out/pp/docs16k.txt 2>&1
python -m pytest tests/test_documents.py -m gpu -x -q > gpurun_out/pp/pytest.txt 2>&1; tail -2 gpurun_out/pp/pytest.txt
cat gpurun_out/pp/docs16k.txt gpurun_out/pp/docs1m.txt
