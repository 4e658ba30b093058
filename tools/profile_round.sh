#!/bin/bash
# Profiles behind the numbers the design notes and bench.py quote (run on the GPU box from the repository root):
# kernel statistics of the bench command, HBM traffic counters (one pass per counter) and the
# issue / busy counters of the Lorenz-96 energy kernel (two passes of at most eight SQ counters).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pf
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-parity > $OUT/stats.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-parity > $OUT/fetch.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-parity > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq_a --output-format csv -- python3 tools/diag/esde_only.py 512 > $OUT/sq_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b --output-format csv -- python3 tools/diag/esde_only.py 512 > $OUT/sq_b.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
for d in fetch write sq_a sq_b; do python3 tools/diag/pmc_summary.py $OUT/$d > $OUT/$d.summary.txt; done
echo profiles done
