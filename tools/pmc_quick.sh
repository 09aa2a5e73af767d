# SQ counters + HBM traffic + kernel trace of the Cornell bench workload (one_dispatch, 256 spp); results under gpurun_out/
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--workload cornell --spp 256 --reps 4"
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES -d $R/gpurun_out/pmc_${tag}_sq1 -o g --output-format csv -- python3 $R/tools/one_dispatch.py $A > $R/gpurun_out/pmc_${tag}_sq1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_VALU_TRANS -d $R/gpurun_out/pmc_${tag}_sq2 -o g --output-format csv -- python3 $R/tools/one_dispatch.py $A > $R/gpurun_out/pmc_${tag}_sq2.log 2>&1 || true
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_fetch -o g --output-format csv -- python3 $R/tools/one_dispatch.py $A > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_write -o g --output-format csv -- python3 $R/tools/one_dispatch.py $A > $R/gpurun_out/pmc_${tag}_write.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -o g --output-format csv -- python3 $R/bench.py --lean --steps 10 > $R/gpurun_out/prof_${tag}.log 2>&1
