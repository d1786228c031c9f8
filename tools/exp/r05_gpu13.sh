mkdir -p gpurun_out/s8
timeout 600 python -m pytest tests/test_gpu_fit_bracketing.py tests/test_gpu_speculative_fits.py tests/test_gpu_lazy_bandwidth.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -5 > gpurun_out/s8/pytest_a.txt
cat gpurun_out/s8/pytest_a.txt
python tools/exp/lcv_f32_check.py > gpurun_out/s8/check.txt 2>&1; grep "8192\|4096\|differ" gpurun_out/s8/check.txt | cut -c1-210
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s8/bench_config3.json 2> gpurun_out/s8/bench_config3.err
python bench.py --no-cpu-baseline > gpurun_out/s8/bench.json 2> gpurun_out/s8/bench.err
python - <<'PY'
import json
for f in ("bench_config3","bench"):
    d=json.loads(open(f"gpurun_out/s8/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["roofline"]["kernel_ms_per_step"].items()}, d["roofline_valu"].get("lcv_evals_per_step"), d["roofline_valu"].get("lcv_evals_f32_per_step"), d["roofline_valu"]["frac"], d["roofline_valu"].get("frac_fp64_only"))
    if "north_star_10k" in d: print(d["north_star_10k"]["ms_per_step"], d["north_star_10k"]["roofline_valu"])
PY
