R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm_pp" -d $OUT/pmc_q8 -o pmc -- python $R/tools/pmc_micro.py q8 0 3 > /dev/null 2> $OUT/pmc_q8.err
DB=$(ls $OUT/pmc_q8/*/*.db $OUT/pmc_q8/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_pmc.py $DB > $OUT/r3_pmc_q8.md
rocprofv3 --kernel-trace --stats --kernel-include-regex "gemm_pp" -d $OUT/tr_q8 -o tr -- python $R/tools/pmc_micro.py q8 0 3 > /dev/null 2>> $OUT/pmc_q8.err
DB2=$(ls $OUT/tr_q8/*/*.db $OUT/tr_q8/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB2 >> $OUT/r3_pmc_q8.md
rm -rf $OUT/pmc_q8 $OUT/tr_q8
cat $OUT/r3_pmc_q8.md | cut -c1-200
