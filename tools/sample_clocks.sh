#!/bin/bash
# sample the shader clock / power while the step-proof loop runs (is the integer-VALU load clock-limited?)
python tools/soak.py 1500 > gpurun_out/soak_clk.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4
  sleep 0.7
done
wait $PID
tail -1 gpurun_out/soak_clk.log
