import sys, os, time; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch, json
import emergent_multiagent_strategies_amd as fa
E,G,A,T=4096,3,3,128; N=G+A
fused = sys.argv[1] == "1"
eng=fa.BatchedFortAttack(E,G,A,100,base_seed=0); st=fa.JointRolloutStorage(T,E,N,device="cuda"); eng.bind_storage(st)
g=torch.Generator(device="cuda").manual_seed(1234)
st.actions.copy_(torch.randint(0,8,st.actions.shape,device="cuda",generator=g)); st.value_preds.copy_(torch.randn(st.value_preds.shape,device="cuda",generator=g))
adv=torch.empty((T,E,N,1),device="cuda"); eng.collect_reset()
for _ in range(60):
    if fused: eng.collect_rollout_gae_normalize(0.99,0.95,out=adv)
    else:
        eng.collect_rollout(0,T); eng.gae_normalize(0.99,0.95,out=adv)
torch.cuda.synchronize()
