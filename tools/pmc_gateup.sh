#!/bin/bash
# Memory-side traffic of the bench's dominant kernel (gate/up projection at M = 3072, tile-major weights): two separate
# counter passes, each a run of its own.  Run from the repository root on the GPU box.
set -e
export TMPDIR=/tmp
RUN="python3 tools/gemm_one.py 24576 4096 3 6 3072 0 1"
rm -rf gpurun_out/pmc_gu_rd gpurun_out/pmc_gu_wr
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/pmc_gu_rd -o p -- $RUN
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/pmc_gu_wr -o p -- $RUN
python3 tools/pmc_traffic.py gpurun_out/pmc_gu_rd gpurun_out/pmc_gu_wr gemm_bf16_stream_kernel gpurun_out/r3_pmc_gateup_traffic.json 302e6 \
  "gemm_bf16_stream_kernel<6,SWIGLU> (gate/up, M=3072 N=24576 K=4096, tile-major weights: the bench's dominant kernel)" \
  "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- $RUN ; second pass --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
