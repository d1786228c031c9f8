mkdir -p gpurun_out/s7
timeout 600 python -m pytest tests/test_gpu_fit_bracketing.py tests/test_gpu_speculative_fits.py tests/test_gpu_lazy_bandwidth.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -5 > gpurun_out/s7/pytest_a.txt
cat gpurun_out/s7/pytest_a.txt
for c in 3 4 5; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s7/bench_config$c.json 2> gpurun_out/s7/bench_config$c.err
  NBP_FIT_F64=1 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s7/bench_config${c}_f64.json 2> gpurun_out/s7/bench_config${c}_f64.err
done
python tools/exp/kd_time.py > gpurun_out/s7/kd_time.txt 2>&1
python - <<'PY'
import json
for c in (3,4,5):
    for s in ("", "_f64"):
        try:
            d=json.loads(open(f"gpurun_out/s7/bench_config{c}{s}.json").read().strip().splitlines()[-1])
            print(c, s, round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["roofline"]["kernel_ms_per_step"].items()}, d["roofline_valu"].get("lcv_evals_per_step"), d["roofline_valu"].get("lcv_evals_f32_per_step"))
        except Exception as e:
            print(c, s, "ERR", e)
PY
cat gpurun_out/s7/kd_time.txt | tail -6
