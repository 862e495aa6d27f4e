#!/bin/bash
# Runs on the MI355X box (gpurun): the rocprofv3 kernel stats of the bench command, the PMC passes behind `traffic` / the clock
# table, and then the default bench line (which quotes those passes).  usage: scripts/profile_round.sh <tag, e.g. r03a> <commit>
# Every pass is checked: a failed rocprofv3 run stops the script instead of leaving empty or stale artefacts behind.
set -eu -o pipefail
TAG=$1; COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
need() { [ -s "$1" ] || { echo "profile_round: $1 is missing or empty" >&2; exit 1; }; }
first() { local f; f=$(find "$1" -name "$2" | head -1); [ -n "$f" ] || { echo "profile_round: no $2 under $1" >&2; exit 1; }; echo "$f"; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs > $OUT/${TAG}_bench_1M_under_rocprof.json 2> $OUT/rocprof_stats.err
cp "$(first $OUT/stats '*kernel_stats.csv')" $OUT/${TAG}_bench_1M_kernel_stats.csv
# (ONE build in these passes -- a kernel's total over the pass is its traffic per step: the item-graph lambda leg, which builds a
# graph of its own, is left out here and counted in a pass of its own below)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --no-config-legs --no-spmm-leg > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --no-config-legs --no-spmm-leg > /dev/null 2> $OUT/pmc_write.err
first $OUT/pmc_fetch '*counter_collection.csv' > /dev/null; first $OUT/pmc_write '*counter_collection.csv' > /dev/null
python3 scripts/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write 1000000 128 $COMMIT > $OUT/${TAG}_pmc_traffic.json
need $OUT/${TAG}_pmc_traffic.json
# the item-graph lambda (signal_sums_kernel) on the graph as the build returns it: three calls, bytes per launch
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s -- python3 scripts/run_signals_once.py 1000000 asbuilt 3 > /dev/null 2> $OUT/pmc_fetch_s.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s -- python3 scripts/run_signals_once.py 1000000 asbuilt 3 > /dev/null 2> $OUT/pmc_write_s.err
python3 scripts/pmc_traffic.py $OUT/pmc_fetch_s $OUT/pmc_write_s 1000000 128 $COMMIT > $OUT/${TAG}_pmc_traffic_spmm.json
need $OUT/${TAG}_pmc_traffic_spmm.json
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_clock -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs > /dev/null 2> $OUT/pmc_clock.err
python3 scripts/pmc_clock.py $OUT/pmc_clock > $OUT/${TAG}_pmc_clock.txt 2>&1
need $OUT/${TAG}_pmc_clock.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_s $OUT/pmc_write_s $OUT/pmc_clock
# the default bench line LAST, with this round's PMC passes in place (profiles/ of this box's copy of the repository): its
# `traffic` then comes from the sources it runs from and `traffic_stale` is false
cp $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_traffic_spmm.json profiles/
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_1M_default.json 2> $OUT/bench_default.err || { tail -c 2000 $OUT/bench_default.err; exit 1; }
need $OUT/${TAG}_bench_1M_default.json
cp bench_detail.json $OUT/${TAG}_bench_1M_detail.json
head -c 1500 $OUT/${TAG}_bench_1M_default.json; echo; head -12 $OUT/${TAG}_bench_1M_kernel_stats.csv | cut -c1-160; cat $OUT/${TAG}_pmc_clock.txt | cut -c1-200 | head
