show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],4))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_collector.py -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 --no-cpu-baseline | show
echo E=2560; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --envs 2560 | show
echo 5v5; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --guards 5 --attackers 5 | show
