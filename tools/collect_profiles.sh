#!/bin/bash
# The round's evidence in one GPU call: plain bench line, kernel trace of the same command (+ timed-region summary +
# per-kernel stats), three PMC passes (HBM traffic), the other configs.  Output under gpurun_out/$TAG; copy what is
# to be judged into profiles/.   Usage (on the GPU box): tools/collect_profiles.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-10k > $O/bench_under_rocprof.json 2> $O/trace.err
python $R/tools/summarize_trace.py $O/trace 5 > $O/bench_timed_region_summary.txt
python $R/tools/kernel_stats.py $O/trace > $O/bench_kernel_stats.csv
for p in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $p -d $O/pmc/$p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass > /dev/null 2> $O/pmc_$p.err; done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc/TCC -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass > /dev/null 2> $O/pmc_TCC.err
python $R/tools/pmc_traffic.py $O/pmc 5 200 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
for c in 3 4 5; do python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err; done
python $R/tools/lcv_bench.py > $O/lcv_microbench.txt 2>/dev/null
# SQ counters of the chip-filling bandwidth fit (two passes of 8 counters)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/sq/p1 -- python $R/tools/lcv_bench.py 200 2048 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU -d $O/sq/p2 -- python $R/tools/lcv_bench.py 200 2048 > /dev/null 2>&1
python $R/tools/pmc_sq.py $O/sq nbp_bandwidth > $O/lcv_sq_counters.txt 2>&1
# SQ counters of a chip-filling batch of proposals / of products (975 of each: one tree level of config 2)
for k in prop prod; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/sq_$k/p1 -- python $R/tools/exp/${k}_batch.py 975 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq_$k/p2 -- python $R/tools/exp/${k}_batch.py 975 > /dev/null 2>&1
done
{ python $R/tools/exp/prop_batch.py 975; python $R/tools/exp/prop_batch.py 1; python $R/tools/pmc_sq.py $O/sq_prop nbp_proposal; } > $O/proposal_sq_counters.txt 2>/dev/null
{ python $R/tools/exp/prod_batch.py 975 2; python $R/tools/exp/prod_batch.py 1 2; python $R/tools/pmc_sq.py $O/sq_prod nbp_product; } > $O/product_sq_counters.txt 2>/dev/null
# what a stream of independent v_fma_f64 reaches on this part (built here if the binary did not travel)
[ -x $R/tools/exp/dp_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $R/tools/exp/dp_rate.hip -o $R/tools/exp/dp_rate 2>/dev/null
$R/tools/exp/dp_rate > $O/dp_rate.txt 2>&1
rm -rf $O/trace/*/*.db.tmp
du -sh $O
