#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pcs
mkdir -p $OUT
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --kernel-trace --output-format csv -d $OUT/ht -- python3 $R/tools/sweep.py --envs 4096 --T 200 --reps 200 > $OUT/ht.log 2>&1
echo "host_trap rc=$?"; tail -5 $OUT/ht.log
ls -R $OUT | head -30
find $OUT -name "*pc_sampling*" | head; for f in $(find $OUT -name "*pc_sampling*csv" | head -2); do head -5 $f; wc -l $f; done
