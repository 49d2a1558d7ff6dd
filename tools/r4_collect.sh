#!/bin/bash
# copies what a measurement round (tools/r4_final.sh TAG) left in gpurun_out/TAG into profiles/ under
# the names DESIGN.md cites: tools/r4_collect.sh TAG PREFIX   (e.g. r04y r04z)
set -u
SRC=gpurun_out/$1; P=profiles/$2
for f in $SRC/bench_*.json; do cp $f ${P}_$(basename $f); done
for f in $SRC/*_kernel_stats.csv $SRC/*_timeline.txt; do cp $f ${P}_$(basename $f); done
cp $SRC/pmc_traffic_c3.json ${P}_pmc_traffic_c3.json
cp $SRC/pmc_traffic_c3dist.json ${P}_pmc_traffic_c3_forcedist.json
cp $SRC/pmc_onesweep_keys.json ${P}_pmc_onesweep_keys.json
cp $SRC/pmc_FETCH_SIZE_sortbench_keys.csv ${P}_pmc_fetch_sortbench_keys_1e8.csv
cp $SRC/pmc_WRITE_SIZE_sortbench_keys.csv ${P}_pmc_write_sortbench_keys_1e8.csv
cp $SRC/sort_bench_keys.txt ${P}_sort_bench.txt
grep -E "passed|failed" $SRC/pytest.log | tail -1 > ${P}_pytest_summary.txt
ls profiles | grep "^$2" | wc -l
