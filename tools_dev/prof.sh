#!/bin/bash
# usage (GPU box): tools_dev/prof.sh <tag> <bench args...>
# rocprofv3 kernel-trace stats of `python3 bench.py <args>` + separate PMC passes (never combined
# with trace domains). Writes gpurun_out/prof_<tag>/summary.txt (+ the raw CSVs).
# PROF_STEPS / PROF_WARMUP (default 5 / 1) set the launches per pass; PROF_EXTRA_ARGS (default --no-extras: only the matrix launches of the all-pairs kernels); PROF_PASSES="trace pmc1 pmc2 pmc3 pmc4 pmc5" selects passes.
set -o pipefail
tag=$1; shift
R=$PWD
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
BENCH_ARGS="--cpu-seconds 0 --steps ${PROF_STEPS:-5} --warmup ${PROF_WARMUP:-1} ${PROF_EXTRA_ARGS---no-extras} --detail prof_detail.json $*"
PASSES=${PROF_PASSES:-trace pmc1 pmc2 pmc3 pmc4 pmc5}
run() { name=$1; shift; case " $PASSES " in *" $name "*) ;; *) return 0;; esac
        (cd /tmp && rocprofv3 "$@" -d $out/$name --output-format csv -- python3 $R/bench.py $BENCH_ARGS > $out/$name.log 2>&1) || { tail -5 $out/$name.log; return 1; }
        echo "pass $name done"; }
run trace --kernel-trace --stats || exit 1
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
run pmc2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU || exit 1
run pmc3 --pmc GRBM_GUI_ACTIVE || exit 1
run pmc4 --pmc FETCH_SIZE || exit 1
run pmc5 --pmc WRITE_SIZE || exit 1
python3 $R/tools_dev/prof_summary.py $out "python3 bench.py $BENCH_ARGS" | tee $out/summary.txt
