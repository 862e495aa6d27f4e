#!/bin/bash
# Runs on the MI355X box (gpurun): the default bench line, the rocprofv3 kernel stats of the same command and the PMC
# passes behind `traffic_recorded` / the clock table.  usage: scripts/profile_round.sh <tag, e.g. r02a> <commit>
set -u
TAG=$1; COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/${TAG}_bench_1M_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/${TAG}_bench_1M_under_rocprof.json 2> $OUT/rocprof_stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_1M_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > /dev/null 2> $OUT/pmc_write.err
python3 scripts/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write 1000000 128 $COMMIT > $OUT/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_clock -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $OUT/pmc_clock.err
python3 scripts/pmc_clock.py $OUT/pmc_clock > $OUT/${TAG}_pmc_clock.txt 2>&1
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_clock
head -c 1500 $OUT/${TAG}_bench_1M_default.json; echo; head -12 $OUT/${TAG}_bench_1M_kernel_stats.csv | cut -c1-160; cat $OUT/${TAG}_pmc_clock.txt | cut -c1-200 | head
