#!/bin/bash
# Round-2 profile collection (run on the GPU box through gpurun): kernel-trace statistics of the default bench, then SEPARATE --pmc
# passes (FETCH_SIZE / WRITE_SIZE / MFMA-busy) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Summaries are copied to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- $B > gpurun_out/r02_stats.json 2> gpurun_out/r02_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_fetch -- $B > /dev/null 2> gpurun_out/r02_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_write -- $B > /dev/null 2> gpurun_out/r02_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02_mfma -- $B > /dev/null 2> gpurun_out/r02_mfma.err
ls gpurun_out/r02_stats/*/ gpurun_out/r02_fetch/*/ gpurun_out/r02_write/*/ gpurun_out/r02_mfma/*/
