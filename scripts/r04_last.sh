# usage (GPU box): bash scripts/r04_last.sh <tag> -- the round's evidence run (scripts/r04_final.sh, changed-path tests only)
# and, behind it, documents through a BPE model that is not word-wise (bpe1k_noesc) next to the reference's thread sweep.
TAG=${1:-r04d}
SKIP_TESTS=1 bash scripts/r04_final.sh $TAG
cd "$GRAFT_REPO_ROOT"
( timeout 150 python scripts/docs_rate.py --model bpe1k_noesc --docs 2048 --bytes 16384 --steps 2 --cpu-seconds 3 ) > gpurun_out/$TAG/docs_rate_bpe1k_noesc.json 2> gpurun_out/$TAG/docs_rate_bpe1k_noesc.err
tail -c 600 gpurun_out/$TAG/docs_rate_bpe1k_noesc.json
