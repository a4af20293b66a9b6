set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3pmc
mkdir -p $O
cd $R
timeout -k 10 60 rocprofv3 -L > $O/counters.txt 2>&1 || true
B="python bench.py --no-cpu-baseline --no-decode --steps 4 --warmup 2 --sweep-ms 1,32,128,1024,4096"
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -o r3 -- $B > $O/sq1.json 2> $O/sq1.err
echo pass1 done
timeout -s KILL 170 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum --output-format csv -d $O/ta1 -o r3 -- $B > $O/ta1.json 2> $O/ta1.err
echo pass2 done
