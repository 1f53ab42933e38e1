show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_us'],2))"; }
for E in 2560; do
echo "E=$E base"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-collector --envs $E | show
for k in 1 8 32 57 59 123; do echo "E=$E abl$k"; FA_LIB_OVERRIDE=exp_libs/libfa_abl$k.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-collector --envs $E | show; done
done
