#!/bin/bash
# long randomised runs of both fuzzers with fresh seeds (GPU box, by hand): bash tools/experiments/long_fuzz.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
rc=0
for seed in 12 13 14 15; do
  timeout -k 10 500 python3 tests/fuzz_api.py 350 $seed > gpurun_out/fuzz_s$seed.log 2>&1 || rc=1
  echo "fuzz_api seed $seed: $(tail -1 gpurun_out/fuzz_s$seed.log | cut -c1-250)"
  timeout -k 10 300 python3 tests/soak.py --fuzz 500 --seed $seed > gpurun_out/soakfuzz_s$seed.log 2>&1 || rc=1
  echo "soak fuzz seed $seed: $(tail -1 gpurun_out/soakfuzz_s$seed.log | cut -c1-250)"
done
exit $rc
