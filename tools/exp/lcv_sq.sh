#!/bin/bash
# lcv_sq.sh N FITS: SQ counters of a chip-filling batch of bandwidth fits at N particles (two passes of eight counters)
N=${1:-300}; F=${2:-4096}
R=${GRAFT_REPO_ROOT:-$PWD}
O=/tmp/lcv_sq_$N; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/p1 -- python $R/tools/lcv_bench.py $N $F > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LDS_ADDR_CONFLICT -d $O/p2 -- python $R/tools/lcv_bench.py $N $F > /dev/null 2>&1
python $R/tools/lcv_bench.py $N $F 2>/dev/null
python $R/tools/pmc_sq.py $O nbp_bandwidth 2>/dev/null | grep -v "^  note"
