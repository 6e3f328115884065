#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the
# headline bench configuration.  Summaries are post-processed by tools/summarise_profile.py
# into profiles/.  Counter passes are kept apart from --kernel-trace/--stats as the pool requires.
set -o pipefail
TAG=${1:-r02}
shift || true
BENCH_ARGS=${*:---steps 1000 --warmup 200 --no-cpu-baseline --no-extras}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp || exit 1
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
echo "== kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $BENCH_ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
echo "== pmc sq1"; rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py $BENCH_ARGS > /dev/null 2> $OUT/pmc_sq1.err || { tail -5 $OUT/pmc_sq1.err; exit 1; }
echo "== pmc sq2"; rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $BENCH_ARGS > /dev/null 2> $OUT/pmc_sq2.err || { tail -5 $OUT/pmc_sq2.err; exit 1; }
echo "== pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $BENCH_ARGS > /dev/null 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
echo "== pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $BENCH_ARGS > /dev/null 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
if [ -n "$PROFILE_MFMA" ]; then
echo "== pmc mfma"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py $BENCH_ARGS > /dev/null 2> $OUT/pmc_mfma.err || { tail -5 $OUT/pmc_mfma.err; }
fi
find $OUT -name '*.csv' | head -40
du -sh $OUT
