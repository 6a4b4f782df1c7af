#!/bin/bash
# Soak of the randomised parity tests on the GPU box: more seeds and trials than the default suite.
#   exact mode   tests/test_fuzz_gpu.py with GRAIL_FUZZ_EXTRA more seeds (bit-identical to the oracle)
#   fast mode    the fuzz tests of tests/test_fast_gpu.py (lane kernels; time-split kernels; batch invariance), new seeds
# usage (through gpurun): tools/fuzz_soak.sh [extra_exact_seeds [fast_seeds [trials_per_seed]]]
extra=${1:-24}; seeds=${2:-8}; trials=${3:-12}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$root" || exit 1
echo "# exact mode, $extra extra seeds per test (bit-identical to the oracle, every lane mapping)"
GRAIL_FUZZ_EXTRA=$extra python3 -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -1
echo "# fast mode, $seeds seeds x $trials trials: worst |fast - oracle| relative to max(1, peak), contract 64 * 2^-23"
for s in $(seq 1 $seeds); do
  GRAIL_FAST_FUZZ_SEED=$((7000 + s)) GRAIL_FAST_FUZZ_TRIALS=$trials python3 -m pytest tests/test_fast_gpu.py -m gpu -q -s \
      -k "fuzz_on_random_voice_tables or invariance_fuzz" 2>&1 | grep -E "fuzz: worst|time-split fuzz|passed|failed|Error|assert" | tr '\n' ' '
  echo " (seed $((7000 + s)))"
done
