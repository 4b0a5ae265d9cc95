cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python3 profiles/r03_mb_pmc.py"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pm1 -- $P > /dev/null 2> gpurun_out/pm1.err &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pm2 -- $P > /dev/null 2> gpurun_out/pm2.err &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pm3 -- $P > /dev/null 2> gpurun_out/pm3.err
python3 profiles/r03_pmc_sum.py gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3
tail -2 gpurun_out/pm1.err gpurun_out/pm2.err gpurun_out/pm3.err | grep -i "error\|fail" | head
rm -rf gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3
