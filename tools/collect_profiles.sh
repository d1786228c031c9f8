#!/bin/bash
# The round's evidence in one GPU call: plain bench line, kernel trace of the same command (+ timed-region summary,
# per-kernel stats, the launches of one solve and of graph initialisation in order), PMC passes (HBM traffic) for the
# default launch form and for the fused update kernel, the other configs, micro-benchmarks and SQ counters.
# Output under gpurun_out/$TAG; copy what is to be judged into profiles/.   Usage (on the GPU box): tools/collect_profiles.sh r04
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
python $R/bench.py --fused-min 256 --no-cpu-baseline --no-10k > $O/bench_fused.json 2> $O/bench_fused.err
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass > $O/bench_under_rocprof.json 2> $O/trace.err
python $R/tools/summarize_trace.py $O/trace 5 > $O/bench_timed_region_summary.txt
python $R/tools/kernel_stats.py $O/trace > $O/bench_kernel_stats.csv
python $R/tools/stage_timeline.py $O/trace > $O/solve_launches_in_order.txt
python $R/tools/init_timeline.py $O/trace > $O/graph_init_launches.txt
rocprofv3 --kernel-trace -d $O/trace4 -- python $R/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass > /dev/null 2> $O/trace4.err
python $R/tools/init_timeline.py $O/trace4 >> $O/graph_init_launches.txt
rm -rf $O/trace4
cd $R
tools/pmc_quick.sh $TAG/pmc_default NBP_X=1 > $O/pmc_traffic.txt 2>&1
tools/pmc_quick.sh $TAG/pmc_fused NBP_FUSED_MIN=256 > $O/pmc_traffic_fused.txt 2>&1
cd /tmp
for c in 3 4 5; do
  python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err
  rocprofv3 --kernel-trace --stats -d $O/trace_c$c -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > /dev/null 2> $O/trace_c$c.err
  python $R/tools/summarize_trace.py $O/trace_c$c 3 > $O/bench_config${c}_timed_region_summary.txt
  python $R/tools/kernel_stats.py $O/trace_c$c > $O/bench_config${c}_kernel_stats.csv
  rm -rf $O/trace_c$c
done
python $R/tools/lcv_bench.py > $O/lcv_microbench.txt 2>/dev/null
# SQ counters of the chip-filling bandwidth fit (two passes of 8 counters)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/sq/p1 -- python $R/tools/lcv_bench.py 200 2048 shipped > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU -d $O/sq/p2 -- python $R/tools/lcv_bench.py 200 2048 shipped > /dev/null 2>&1
python $R/tools/pmc_sq.py $O/sq nbp_bandwidth > $O/lcv_sq_counters.txt 2>&1
# SQ counters of a chip-filling batch of proposals / of products (975 of each: one tree level of config 2)
for k in prop prod; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/sq_$k/p1 -- python $R/tools/exp/${k}_batch.py 975 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq_$k/p2 -- python $R/tools/exp/${k}_batch.py 975 > /dev/null 2>&1
done
{ python $R/tools/exp/prop_batch.py 975; python $R/tools/exp/prop_batch.py 1; python $R/tools/pmc_sq.py $O/sq_prop nbp_proposal; } > $O/proposal_sq_counters.txt 2>/dev/null
{ python $R/tools/exp/prod_batch.py 975 2; python $R/tools/exp/prod_batch.py 1 2; python $R/tools/pmc_sq.py $O/sq_prod nbp_product; } > $O/product_sq_counters.txt 2>/dev/null
# single-precision bracketing of the bandwidth searches: bit-equal bandwidths and timings with / without (NBP_FIT_F64=1), every
# evaluation against the host's and the bound, what a pair costs in the candidate loop forms, small launches, the configs either way
bash $R/tools/exp/fit_bracketing_record.sh > $O/fit_bracketing.txt 2>&1
# debug build (tools/libnbp_dbg.so, -DNBP_PHASE_TIMING): where one evaluation of a fit and one fused workgroup spend their
# time, and how busy the lanes of the per-particle searches are
if [ -f $R/tools/libnbp_dbg.so ]; then
  NBP_NO_SPECULATIVE_FITS=1 NBP_FIT_F64=1 python $R/tools/lcv_phase_timing.py > $O/lcv_phase_timing.txt 2>/dev/null
  python $R/tools/product_phase_timing.py > $O/product_phase_timing.txt 2>/dev/null
  python $R/tools/exp/nm_lane_util.py > $O/search_lane_utilisation.txt 2>/dev/null
  for a in "488 2" "975 2" "4000 2"; do python $R/tools/exp/fused_phase.py $a 2>/dev/null; done > $O/fused_phase_timing.txt
fi
# several ranks sharing the one GPU (gloo, host-staged exchange): every leg of tests/test_gpu_bench.py three times, the posterior sha of every rank
(cd $R && tools/exp/loop_multirank.sh 3 > /dev/null 2>&1; cp gpurun_out/mr_loop/summary.txt $O/multirank_shared_gpu.txt)
# the clique seam from plain C: one call per clique (1 / 4 / 16 callers), a batched call per tree level, queued batches
(cd $R && NBP_PLAN_CACHE_STATS=1 bash tools/exp/clique_seam_rate.sh > $O/clique_seam_rate.txt 2>&1; bash tools/exp/clique_seam_phases.sh > $O/clique_seam_phases.txt 2>&1; NBP_PLAN_CACHE=0 bash tools/exp/clique_seam_rate.sh > $O/clique_seam_rate_no_plan_cache.txt 2>&1)
rm -rf $O/trace $O/sq $O/sq_prop $O/sq_prod
du -sh $O
