#!/bin/bash
# tools/ubench/mfma_power.sh : the two MFMA shapes (and all-zero operands), with power / clock sampled by rocm-smi meanwhile
cd "$(dirname "$0")"
for args in "0 6" "1 6" "0 6 z" "1 6 z"; do
  ./mfma_power $args &
  pid=$!
  sleep 2
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr "\n" " "; echo; done
  wait $pid
done
