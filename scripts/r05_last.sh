# usage (GPU box): bash scripts/r05_last.sh <tag>  -- what was added after the round's evidence run (scripts/r05_final.sh): the GPU
# twin of the universal-word-form test, and the PMC passes behind `long_piece_model.roofline.traffic` (uni32k_w16)
TAG=${1:-r05last}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 400 python -m pytest tests/test_word_form.py tests/test_emu.py -m gpu -x -q -k "not_plain_ascii or nul_and_control" ) > $O/pytest_gpu_new.txt 2>&1; tail -4 $O/pytest_gpu_new.txt
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k_w16 > $O/pmc_traffic_w16.log 2>&1; tail -12 $O/pmc_traffic_w16.log | cut -c1-200
# ... and the N > 1 control flow of bench.py (process group over RCCL, both gather algorithms, 16 CUs reserved, the watchdog) with ONE rank
SPMX_BENCH_ONE_RANK_GATHER=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-second-model --no-side-configs --no-cpu-baseline > $O/bench_one_rank_gather.json 2> $O/bench_one_rank_gather.err; tail -c 1500 $O/bench_one_rank_gather.json; tail -3 $O/bench_one_rank_gather.err
