import time, torch, sys
sys.path.insert(0, "/root/repo")
import emergent_multiagent_strategies_amd as fa
torch.manual_seed(0)
eng = fa.BatchedFortAttack(4096, 3, 3, 100, track_counters=False)
L = fa.BatchedLearner(eng, num_steps=128, use_graph=True)
L.reset(); L.collect(); 
torch.cuda.synchronize(); t0=time.perf_counter(); L.update(); torch.cuda.synchronize(); t1=time.perf_counter()
print("update s", t1-t0)
from torch.profiler import profile, ProfilerActivity
L.ppo_epoch=1; L.num_mini_batch=32
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    L.update(train_guards_only=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
