#!/bin/bash
# PMC pass for the MAAC-R scorer (GPU box): MFMA pipe utilisation and wave-cycle breakdown.
set -o pipefail
OUT=$PWD/gpurun_out/prof_pmi_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 $R/bench.py --reward pmi --steps 200 --warmup 200 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/a.err || tail -3 $OUT/a.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 $R/bench.py --reward pmi --steps 200 --warmup 200 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/b.err || tail -3 $OUT/b.err
find $OUT -name '*counter_collection.csv'
