#!/bin/bash
# run the probe repeatedly in background, sample clocks
(for i in 1 2 3 4 5 6; do tools/microbench/gemm_probe_p > /dev/null; done) &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power\|mclk" | head -4; sleep 1.5; done
wait $PID
echo idle; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power" | head -3
