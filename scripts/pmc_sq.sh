# SQ / TCP / TCC counters of the encode kernels (a TA_* group hung the profiler on this pool and was dropped) (one rocprofv3 pass per counter group; gpurun refuses --pmc with
# other trace domains, so only --kernel-trace is combined).  Usage on the GPU box: [MODEL=..] [CORPUS=docs_16k] bash scripts/pmc_sq.sh [sentences]
N=${1:-2000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_sq
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TD|SQ|TCC)_[A-Z0-9_a-z]+" | sort -u > gpurun_out/pmc_sq/avail.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_CYCLES"; do
  i=$((i+1))
  if [ -n "$GROUPS_MAX" ] && [ $i -gt $GROUPS_MAX ]; then break; fi
  timeout 70 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc_sq/g$i -o pmc -- python bench.py --model ${MODEL:-uni32k} ${CORPUS:+--corpus $CORPUS} --sentences $N --steps 2 --warmup 1 --no-cpu-baseline --no-second-model --no-side-configs > gpurun_out/pmc_sq/g$i.log 2>&1
  echo "group $i rc=$?"
done
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob('gpurun_out/pmc_sq/g*/**/pmc_results.db', recursive=True)):
    db = sqlite3.connect(f)
    try:
        q = ("select k.kernel_name, p.counter_name, count(*), avg(p.value) from pmc_events p join kernels k on p.event_id = k.event_id "
             "where (k.kernel_name like '%Encode%Kernel%' or k.kernel_name like '%UniLong%') group by k.kernel_name, p.counter_name order by 1, 2")
        rows = list(db.execute(q))
    except sqlite3.Error:
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection where (kernel_name like '%Encode%Kernel%' or kernel_name like '%UniLong%') "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        rows = list(db.execute(q))
    for r in rows:
        if r[3] > 0: print("%-46s %-36s n=%d avg=%.5g" % (r[0][:46], r[1], r[2], r[3]))
PY
rm -rf gpurun_out/pmc_sq/g*/
