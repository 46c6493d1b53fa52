#!/bin/bash
# Socket power and shader clock of the down-projection GEMM (M=3072, N=4096, K=12288: the per-tile pipelined kernel)
# looping, for the shipped build and the timing-only ablations of its loop (-DLLADA_GEMM_DEBUG: 1 no DMA after the
# first two K tiles, 2 no LDS fragment reads, 4 no MFMAs, 6 neither reads nor MFMAs); libraries under ab/.
sample() {
  sleep "$1"
  for i in 1 2 3; do
    c=$(rocm-smi --showclocks 2>/dev/null | grep -oE "sclk clock level: \S+ \([0-9]+Mhz\)" | grep -oE "[0-9]+Mhz")
    p=$(amd-smi metric --power 2>/dev/null | grep -oE "SOCKET_POWER: *[0-9]+ W")
    echo "   $c  $p"
    sleep 2
  done
}
for v in "" dbg1 dbg2 dbg4 dbg6; do
  if [ -n "$v" ]; then export LLADA_HIP_LIB=$PWD/ab/libllada_hip_$v.so; else unset LLADA_HIP_LIB; fi
  echo "== build: ${v:-shipped}"
  python3 -u tools/gemm_one.py 4096 12288 0 100000 3072 0 1 > /dev/null 2>&1 &
  P=$!; sample 9; wait $P
done
