#!/bin/bash
# prop_sq.sh KIND B: SQ counters of a batch of B proposals of one kind (lin2 | lin3 | se2 | circ), two passes of eight counters,
# and what they say about the SIMDs: resident waves per SIMD and the share of the SIMD-time in which something issues
KIND=${1:-lin2}; B=${2:-975}
R=${GRAFT_REPO_ROOT:-$PWD}
O=/tmp/prop_sq_$KIND_$B; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/p1 -- python $R/tools/exp/prop_batch.py $B 200 $KIND > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 -- python $R/tools/exp/prop_batch.py $B 200 $KIND > /dev/null 2>&1
python $R/tools/exp/prop_batch.py $B 200 $KIND 2>/dev/null | tee $O/time.txt
python $R/tools/pmc_sq.py $O nbp_proposal > $O/sq.txt 2>/dev/null
python - <<PY
import re
t = float(re.search(r"proposals: ([0-9.]+) us", open("$O/time.txt").read()).group(1))
c = {}
for line in open("$O/sq.txt"):
    m = re.match(r"\s+(SQ_\w+)\s+(\d+)\s+\((\d+) dispatches\)", line)
    if m: c[m.group(1)] = float(m.group(2)) / int(m.group(3))
for ghz in (2.0, 2.4):
    simd_quads = t * 1e-6 * ghz * 1e9 / 4 * 1024  # quad-cycles of all 1024 SIMDs over one launch
    print("  at %.1f GHz: resident waves per SIMD (time average) %.2f; SIMD-time with an instruction issuing %.2f (VALU %.2f); per wave: issuing %.2f, waiting to issue %.2f, waiting for a barrier / a counter %.2f"
          % (ghz, c["SQ_WAVE_CYCLES"] / simd_quads, c["SQ_ACTIVE_INST_ANY"] / simd_quads, c["SQ_ACTIVE_INST_VALU"] / simd_quads,
             c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
print("  VALU instructions per launch %.3g, waves %d" % (c["SQ_INSTS_VALU"], c["SQ_WAVES"]))
PY
