#!/bin/bash
# 20 full-size solves per BASELINE configuration, a different seed each (bench.py --steps 20: nbp_program_reseed between the
# steps), the configuration's acceptance criteria checked after the last, then the profiling pass with the counters.
R=${GRAFT_REPO_ROOT:-$PWD}
echo "bench.py --config C --steps 20 --warmup 1 (one MI355X, the round's final library): 20 full-size solves per BASELINE configuration, a different seed each"
echo "(nbp_program_reseed), the configuration's acceptance criteria checked after the last (bench_support.check_posteriors raises otherwise), then the"
echo "profiling pass with the device-side counters read back"
echo
for c in 2 2p 3 4 5; do
  python $R/bench.py --config $c --steps 20 --warmup 1 --no-cpu-baseline --no-10k 2>/tmp/soak_$c.err | python -c "
import json, sys
try:
    d = json.load(sys.stdin)
    v = d['roofline_valu']
    print('config $c: %.2f ms per solve over 20 solves (a different seed each); %d per-particle searches not converged within 1000 iterations and %d NaN results in the profiled solves; max |posterior mean - truth| over the sampled poses %s; share of particles at the true pose (min, median) %s; fits: %.0f double + %.0f single evaluations per solve' % (d['ms_per_step'], v['nonconverged_solves'], v['nan_results'], d['posterior_max_mean_err'], d['posterior_mode_share_min_median'], v['lcv_evals_per_step'], v['lcv_evals_f32_per_step']))
except Exception as e:
    print('config $c: FAILED', e); print(open('/tmp/soak_$c.err').read()[-1500:])
"
done
