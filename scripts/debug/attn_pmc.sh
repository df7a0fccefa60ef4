# PMC counters of the attention kernel alone (SQ block: 8 slots per pass)
REPO_DIR=$PWD
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  rm -rf gpurun_out/attn_pmc && mkdir -p gpurun_out/attn_pmc
  (cd /tmp && rocprofv3 --pmc $SET --kernel-trace -d $REPO_DIR/gpurun_out/attn_pmc -o a -- python $REPO_DIR/scripts/debug/attn_ab.py 2 > /dev/null 2>&1)
  python - <<'PY'
import sqlite3,glob
db=glob.glob('gpurun_out/attn_pmc/*.db')[0]
con=sqlite3.connect(db); cur=con.cursor()
rows=cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn_fwd2%' group by kernel_name, counter_name, grid_size order by grid_size").fetchall()
for r in rows: print(r[0][:40], r[1], f"{r[2]:.4g}", r[3])
PY
done
