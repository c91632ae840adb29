tag=r03
export TMPDIR=/tmp
tools/prof_stats.sh ${tag} --steps 200 --warmup 20 > /dev/null
tools/prof_pmc.sh ${tag}_fetch "FETCH_SIZE" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_fetch.txt
tools/prof_pmc.sh ${tag}_write "WRITE_SIZE" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_write.txt
tools/prof_pmc.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_sq.txt
python3 tools/make_traffic.py gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt estep_docs_reg_kernel,estep_docs_tiered_kernel $1 > gpurun_out/${tag}_traffic.json
cat gpurun_out/${tag}_kernel_stats.csv | cut -c1-200 | head -8; cat gpurun_out/${tag}_traffic.json | head -12
