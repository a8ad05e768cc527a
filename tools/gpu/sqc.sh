set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/sqc; mkdir -p $OUT
run() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python3 bench.py --steps 5 --warmup 2 --repeats 1 --settle-seconds 0 --no-cpu-baseline --no-general-psf > $OUT/$name.log 2>&1; }
run a SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run b SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES
run c SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAVE_CYCLES
run d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run e SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS
python3 tools/pmc_summary.py $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e > $OUT/sq.txt
find $OUT -name "*.csv" -size +3M -delete
grep -A6 "gmm_screen" $OUT/sq.txt
