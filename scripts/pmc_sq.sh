# SQ stall breakdown of the encode kernels (one rocprofv3 pass per counter group; gpurun refuses --pmc with
# other trace domains, so only --kernel-trace is combined).  Usage on the GPU box: bash scripts/pmc_sq.sh [sentences]
N=${1:-2000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_sq
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc_sq/g$i -o pmc -- python bench.py --sentences $N --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq/g$i.log 2>&1
  echo "group $i rc=$?"
done
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob('gpurun_out/pmc_sq/g*/pmc_results.db')):
    db = sqlite3.connect(f)
    for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%Encode%Kernel%' group by kernel_name, counter_name order by kernel_name, counter_name"):
        if r[3] > 0: print("%-46s %-28s n=%d avg=%.4g" % (r[0][:46], r[1], r[2], r[3]))
PY
