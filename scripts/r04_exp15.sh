# usage (GPU box): bash scripts/r04_exp15.sh -- SQ counters of the two word rounds on uni32k_w16 (2 M sentences): where does the second round's time go?  gpurun_out/r04q/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o pmc -- python bench.py --model uni32k_w16 --sentences 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-second-model --no-side-configs > $O/g$i.log 2>&1
  echo "group $i rc=$?"
done
python - <<'PY' > gpurun_out/r04q/w16_2m_pmc_sq.txt
import sqlite3, glob
print("# rocprofv3 --kernel-trace --pmc <group> -- python bench.py --model uni32k_w16 --sentences 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-second-model --no-side-configs (one pass per group)")
for f in sorted(glob.glob('gpurun_out/r04q/g*/**/pmc_results.db', recursive=True)):
    db = sqlite3.connect(f)
    try:
        q = ("select k.kernel_name, p.counter_name, count(*), avg(p.value) from pmc_events p join kernels k on p.event_id = k.event_id "
             "where k.kernel_name like '%Encode%Kernel%' group by k.kernel_name, p.counter_name order by 1, 2")
        rows = list(db.execute(q))
    except sqlite3.Error:
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%Encode%Kernel%' "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        rows = list(db.execute(q))
    for r in rows:
        if r[3] > 0: print("%-46s %-36s n=%d avg=%.5g" % (r[0][:46], r[1], r[2], r[3]))
PY
rm -rf $O/g*/
cat $O/w16_2m_pmc_sq.txt | grep -i "EncodeWord" | cut -c1-120
