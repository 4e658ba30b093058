#!/bin/bash
# Timings of the configurations other than the headline one and of the bench variants (run on the GPU box from the
# repository root); the outputs are put together into profiles/<tag>_other_configs.txt.
set -e
OUT=gpurun_out/other
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 python3 tools/diag/config_times.py > $OUT/config_times.txt 2> $OUT/config_times.err
timeout -k 10 120 python3 tools/diag/small_scan_times.py 512 4096 > $OUT/small_scans.txt 2>> $OUT/config_times.err
METHOD=RK4 timeout -k 10 120 python3 tools/diag/small_scan_times.py 512 >> $OUT/small_scans.txt 2>> $OUT/config_times.err
timeout -k 10 120 python3 tools/diag/facade_time.py > $OUT/facade.txt 2>&1
timeout -k 10 120 python3 tools/diag/captured_time.py > $OUT/captured.txt 2>&1
timeout -k 10 200 python3 bench.py --mode windows --no-cpu --steps 10 > $OUT/windows_512.json 2> $OUT/windows.err
timeout -k 10 200 python3 bench.py --mode windows --no-cpu --steps 10 --problems 1 > $OUT/windows_1.json 2>> $OUT/windows.err
VGPA_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu --steps 5 > $OUT/force_dist.json 2> $OUT/force_dist.err
echo other configs done
