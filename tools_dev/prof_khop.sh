#!/bin/bash
# usage (GPU box): tools_dev/prof_khop.sh <tag> <khop_time.py shapes...>
# rocprofv3 kernel-trace stats of the k-hop sweep A/B (tools_dev/khop_time.py) + separate PMC passes (GRBM, FETCH, WRITE)
#   -> gpurun_out/prof_<tag>/{stats.txt, by_grid.txt, summary.txt}.  PROF_PASSES="trace" skips the counters.
set -o pipefail
tag=$1; shift
R=$PWD
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
PASSES=${PROF_PASSES:-trace pmc3 pmc4 pmc5}
run() { name=$1; shift; case " $PASSES " in *" $name "*) ;; *) return 0;; esac
        (cd /tmp && rocprofv3 "$@" -d $out/$name --output-format csv -- python3 $R/tools_dev/khop_time.py $SHAPES > $out/$name.log 2>&1) || { tail -5 $out/$name.log; return 1; }
        echo "pass $name done"; }
SHAPES="$*"
run trace --kernel-trace --stats || exit 1
run pmc3 --pmc GRBM_GUI_ACTIVE || exit 1
run pmc4 --pmc FETCH_SIZE || exit 1
run pmc5 --pmc WRITE_SIZE || exit 1
grep "^[a-z]" $out/trace.log | grep -v "^W2\|^E2" | tee $out/stats.txt
python3 $R/tools_dev/trace_by_grid.py $(find $out/trace -name '*kernel_trace.csv' | head -1) | tee $out/by_grid.txt
python3 $R/tools_dev/prof_summary.py $out "python3 tools_dev/khop_time.py $SHAPES" | tee $out/summary.txt
