#!/bin/bash
cd "$(dirname "$0")/../.."
for e in ${EXPS:-6 8 10}; do
  echo "== KAO_LP_SIGEXP=$e"
  KAO_LP_SIGEXP=$e timeout 600 python -m pytest tests/test_gpu_lp.py -m gpu -q -p no:cacheprovider -k "test_lp_trace_matches_the_restatement or test_a_band_whose_slack" -s 2>&1 | grep -E "^E  +(assert|Assertion)|cap \+ 1|passed|failed|^FAILED" | cut -c1-300
done
