# HBM traffic of the encode kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (the TCC
# block cannot hold both; gpurun also refuses --pmc together with other trace domains), on the headline bench
# command.  Usage on the GPU box: bash scripts/pmc_traffic.sh <tag> [sentences]
TAG=${1:-r02}; N=${2:-10000000}; MODEL=${3:-uni32k}; CORPUS=${CORPUS:-synthetic}
KEY=$MODEL; [ "$CORPUS" != synthetic ] && KEY=$MODEL@$CORPUS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  timeout ${PASS_TIMEOUT:-200} rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python bench.py --model $MODEL --corpus $CORPUS --sentences $N --steps 2 --warmup 1 --no-cpu-baseline --no-second-model --no-side-configs > $O/pmc_$C.log 2>&1
  echo "$C rc=$?"
done
python - "$O" "$N" "$KEY" <<'PY'
import sqlite3, glob, sys, json
O, N, MODEL = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sys.path.insert(0, '.')
import bench
SHA = bench.kernel_sources_sha()
rows = []
for f in sorted(glob.glob(O + '/pmc_*/**/pmc_results.db', recursive=True)):
    db = sqlite3.connect(f)
    try:
        q = ("select k.kernel_name, p.counter_name, count(*), avg(p.value), min(p.value), max(p.value) from pmc_events p "
             "join kernels k on p.event_id = k.event_id group by 1, 2")
        rows += list(db.execute(q))
    except sqlite3.Error:
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection group by 1, 2")
        rows += list(db.execute(q))
rows.sort(key=lambda r: (r[1], -r[3]))
with open(O + '/%s_pmc_traffic.txt' % MODEL, 'w') as f:
    f.write("# rocprofv3 --kernel-trace --pmc <C> -- python bench.py --model %s --sentences %d --steps 2 --warmup 1 --no-cpu-baseline --no-second-model --no-side-configs   (one pass per counter; kernel sources %s)\n" % (MODEL, N, SHA))
    f.write("# values are KB per dispatch (rocprofv3 FETCH_SIZE / WRITE_SIZE units)\n")
    f.write("%-62s %-11s %3s %16s %16s %16s\n" % ("kernel", "counter", "n", "avg", "min", "max"))
    for r in rows:
        f.write("%-62s %-11s %3d %16.1f %16.1f %16.1f\n" % (r[0][:62], r[1], r[2], r[3], r[4], r[5]))
agg = {}
for r in rows:
    agg.setdefault(r[0], {})[r[1]] = r[3]
out = {}
for k, v in agg.items():
    if ('Encode' in k or 'UniLong' in k or 'BpeLong' in k) and v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0) > 1000:
        name = k.split('(')[0].replace('void ', '').replace('spmx::', '')
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 16 B/lane reads at half their bytes -> doubled
        out["%s:%d:%s" % (MODEL, N, name)] = {"bytes": int((2 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) * 1024),
                                               "bytes_lower": int((v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) * 1024),
                                               "fetch_kb": v.get('FETCH_SIZE', 0), "write_kb": v.get('WRITE_SIZE', 0), "src_sha": SHA,
                                               "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch, separate --pmc passes"}
out["_note"] = ("per dispatch: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate --pmc passes; FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 tallies the 16 B/lane global loads these kernels issue at half their bytes)")
json.dump(out, open(O + '/pmc_traffic_%s.json' % MODEL, 'w'), indent=1)
print(open(O + '/%s_pmc_traffic.txt' % MODEL).read()[:3000])
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
