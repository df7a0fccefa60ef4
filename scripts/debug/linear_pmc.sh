# PMC counters of the K-resident linear kernel alone (TCC write path + SQ mix); PROF_MODE selects a diagnostic mode
REPO_DIR=$PWD
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/tcc_counters.txt
for SET in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "WRITE_SIZE FETCH_SIZE" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITE_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  rm -rf gpurun_out/lin_pmc && mkdir -p gpurun_out/lin_pmc
  (cd /tmp && PROF=1 rocprofv3 --pmc $SET --kernel-trace -d $REPO_DIR/gpurun_out/lin_pmc -o a -- python $REPO_DIR/scripts/debug/linear_ab.py > /dev/null 2>&1)
  python - <<'PY'
import sqlite3,glob
dbs=glob.glob('gpurun_out/lin_pmc/*.db')
if not dbs: print("no db"); raise SystemExit
con=sqlite3.connect(dbs[0]); cur=con.cursor()
rows=cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%linear_k384%' group by kernel_name, counter_name").fetchall()
for r in rows: print(r[0][:40], r[1], f"{r[2]:.5g}", r[3])
PY
done
rm -rf gpurun_out/lin_pmc
