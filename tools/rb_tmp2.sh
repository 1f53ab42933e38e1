show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],4))"; }
for E in 262144; do
for rep in 1 2; do
echo "E=$E base"; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-collector --envs $E --rollout 64 | show
echo "E=$E c6e4c65"; FA_LIB_OVERRIDE=exp_libs/libfa_c6e4c65.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-collector --envs $E --rollout 64 | show
done; done
