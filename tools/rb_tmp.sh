show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],4))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_collector.py -q -x 2>&1 | tail -3
for p in 2 1; do echo PIPE=$p; FA_PIPE=$p python bench.py --steps 20 --warmup 3 --no-cpu-baseline | show; done
echo E=2560; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --envs 2560 | show
for p in 2; do echo "== PIPE=$p"; FA_PIPE=$p FA_LIB_OVERRIDE=exp_libs/libfa_timing.so python tools/timing_probe.py 4096; done
