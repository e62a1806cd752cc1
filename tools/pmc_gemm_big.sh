# SQ counters of the big-tile GEMM (one group per pass, kernel trace only):  bash tools/pmc_gemm_big.sh   -> gpurun_out/pmc_big/*.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_big; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o r --output-format csv -- python tools/bench_gemm_big.py --M 262144 --one > /dev/null 2> $OUT/err$i.log
  python - $OUT/p$i <<'PY' > $OUT/group$i.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/r_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if "big_gemm" not in k and "gemm_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in acc[k].items(): print("   %-28s %.4g per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
PY
  rm -rf $OUT/p$i
done
cat $OUT/group*.txt
