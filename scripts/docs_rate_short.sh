cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for a in "--docs 8192 --bytes 16384" "--docs 256 --bytes 1048576"; do
python scripts/docs_rate.py $a --cpu-seconds 0.3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['docs'], d['doc_bytes'], '%.1f MB/s' % d['gpu_mb_per_s'], d['kernels_ms'], 'exact', d.get('probe_ids_bit_exact'), 'cpu best %.0f' % d.get('cpu_best_mb_per_s', 0))"
done
