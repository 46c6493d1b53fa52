#!/bin/bash
# Shader clock (and power, where the tools report it) sampled while a workload loops on the GPU:
#   bash tools/clk_probe.sh            -> gate/up GEMM loop, 8-row generate() steps, batch-1 steps, MFMA-only probe
sample() {   # $1 = seconds to wait before sampling, $2 = samples
  sleep "$1"
  for i in $(seq 1 "$2"); do
    rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
    rocm-smi --showpower 2>/dev/null | grep -iE "power.*\(W\)|Socket" | head -2
    amd-smi metric --power --clock 2>/dev/null | grep -iE "SOCKET_POWER|CLK:|GFX_0" | head -4
    sleep 2
  done
}
echo "== gate/up GEMM (M=3072, N=24576, K=4096) looping"
python3 -u tools/gemm_one.py 24576 4096 3 60000 3072 0 1 > /dev/null 2>&1 &
P=$!; sample 10 5; wait $P
echo "== generate() steps, 8 rows x 384 tokens"
python3 -u tools/step_time.py 8 128 256 32 400 > gpurun_out/clk_step8.log 2>&1 &
P=$!; sample 25 5; wait $P; tail -1 gpurun_out/clk_step8.log
echo "== generate() steps, 1 row x 384 tokens"
python3 -u tools/step_time.py 1 128 256 32 1500 > gpurun_out/clk_step1.log 2>&1 &
P=$!; sample 25 5; wait $P; tail -1 gpurun_out/clk_step1.log
