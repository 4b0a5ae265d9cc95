#!/bin/bash
# Round-4 profile collection (run on the GPU box through gpurun): kernel-trace statistics of the default bench, then SEPARATE --pmc
# passes (FETCH_SIZE / WRITE_SIZE / MFMA-busy / vector-memory path) as /opt/skills/guides/MI355X_MICROARCH.md prescribes (no --pmc together
# with sys / runtime traces).  profiles/summarize_r04.py turns the outputs into the committed summaries.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --f32-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_stats -- $B > gpurun_out/r04_stats.json 2> gpurun_out/r04_stats.err &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r04_fetch -- $B > /dev/null 2> gpurun_out/r04_fetch.err &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r04_write -- $B > /dev/null 2> gpurun_out/r04_write.err &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r04_mfma -- $B > /dev/null 2> gpurun_out/r04_mfma.err &&
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r04_ta -- $B > /dev/null 2> gpurun_out/r04_ta.err
ls gpurun_out/r04_stats/*/ gpurun_out/r04_fetch/*/ gpurun_out/r04_write/*/ gpurun_out/r04_mfma/*/ gpurun_out/r04_ta/*/
