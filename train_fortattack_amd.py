#!/usr/bin/env python
"""Training entry point with the reference's flag names (arguments.py) over the batched
MI355X engine: the loop of train_fortattack.py:40-150 / train_fortattack_v2.py (ensemble
attackers with --train-guards-only) with --num-processes envs per GPU instead of one.

    python train_fortattack_amd.py --num-processes 4096 --num-steps 128 --num-frames 50000000 --save-dir run1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_fortattack_amd.py ...

Differences from the reference script: no tensorboard / interactive save-dir prompt; one
line of JSON per update on rank 0; `--num-guards/--num-attackers` exist (the reference is
hard-wired to 5v5); the env RNG is seeded per env (`seed + global env index`).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

import emergent_multiagent_strategies_amd as fa


def get_args():
    p = argparse.ArgumentParser(description="RL")
    p.add_argument("--num-guards", type=int, default=5)
    p.add_argument("--num-attackers", type=int, default=5)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--num-processes", type=int, default=4096, help="envs per GPU (reference: 1)")
    p.add_argument("--num-steps", type=int, default=128, help="rollout length (reference default 1000)")
    p.add_argument("--num-env-steps", type=int, default=100, help="max steps per episode")
    p.add_argument("--num-frames", type=int, default=int(50e6))
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--tau", type=float, default=0.95)
    p.add_argument("--entropy-coef", type=float, default=0.01)
    p.add_argument("--value-loss-coef", type=float, default=0.5)
    p.add_argument("--max-grad-norm", type=float, default=0.5)
    p.add_argument("--ppo-epoch", type=int, default=4)
    p.add_argument("--num-mini-batch", type=int, default=32)
    p.add_argument("--clip-param", type=float, default=0.2)
    p.add_argument("--no-clipped-value-loss", action="store_true")
    p.add_argument("--save-dir", default="tmp")
    p.add_argument("--reference-sampling", action="store_true",
                   help="draw the PPO minibatches as the reference does (torch.randperm on the CPU generator per epoch and team, "
                        "rlcore/algo/ppo.py:213) instead of on the device")
    p.add_argument("--save-interval", type=int, default=10)
    p.add_argument("--log-interval", type=int, default=1)
    p.add_argument("--continue-training", action="store_true")
    p.add_argument("--load-dir", default=None, help="checkpoint file to continue from")
    p.add_argument("--ckpt", type=int, default=0, help="update index of --load-dir; training resumes at ckpt+1 "
                                                         "(train_fortattack.py:42)")
    p.add_argument("--train-guards-only", action="store_true")
    p.add_argument("--attacker-load-dir", default="tmp")
    p.add_argument("-l", "--attacker-ckpts", nargs="+", type=int, default=[220, 650, 1240, 1600, 2520])
    p.add_argument("--guard-load-dir", default=None, help="pretrained guard checkpoint file (--pretrained-guard)")
    p.add_argument("--no-graph", action="store_true", help="do not replay the per-step sequence from hipGraphs")
    p.add_argument("--no-pin", action="store_true", help="several ranks: do not pin the process to its slice of the GPU-local cores")
    return p.parse_args()


def main():
    args = get_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world > 1 and not args.no_pin:
        # every update ends in cross-rank exchanges: each rank's host thread stays on its own slice of its GPU's NUMA-local cores
        from emergent_multiagent_strategies_amd import dist as fa_dist
        box = [None] * world
        dist.all_gather_object(box, (rank, fa_dist.gpu_local_cpulist(local_rank)[0]))
        pin = fa_dist.pin_rank_to_gpu_local_cpus(local_rank, rank, world, [l for _, l in sorted(box)])
        print("rank %d -> cuda:%d, pinned to %s cpus (%s)" % (rank, local_rank, pin.get("pinned_to"), pin.get("source")), file=sys.stderr, flush=True)
    torch.manual_seed(args.seed)                       # same initial policies on every rank
    E = args.num_processes
    eng = fa.BatchedFortAttack(E, args.num_guards, args.num_attackers, args.num_env_steps, base_seed=args.seed,
                               env_offset=rank * E, device=local_rank)
    L = fa.BatchedLearner(eng, num_steps=args.num_steps, lr=args.lr, clip_param=args.clip_param,
                          ppo_epoch=args.ppo_epoch, num_mini_batch=args.num_mini_batch,
                          value_loss_coef=args.value_loss_coef, entropy_coef=args.entropy_coef,
                          max_grad_norm=args.max_grad_norm, gamma=args.gamma, tau=args.tau,
                          clipped_value_loss=not args.no_clipped_value_loss, use_graph=not args.no_graph,
                          sample_seed=args.seed + 1, reference_sampling=args.reference_sampling)
    # Action sampling (fused policy kernel): Philox keyed by (sample_seed; rollout counter, step, GLOBAL env index,
    # agent) -- the same seed on every rank, the env index makes the shards differ.  torch's generator only draws
    # the minibatch permutations (and the samples of the PyTorch policy fallback): different per rank.
    torch.manual_seed(args.seed + 1 + rank)
    if args.continue_training:
        if not args.load_dir:
            raise SystemExit("--continue-training needs --load-dir (the checkpoint file to resume from)")
        L.load(args.load_dir)                          # (restores the sampling position when the checkpoint has it)
        if int(L._rollout_counter.item()) == 0:        # a reference checkpoint: continue past the rollouts already made
            L._rollout_counter.fill_(args.ckpt + 1)
    if args.guard_load_dir:
        L.policies[0].load_state_dict(torch.load(args.guard_load_dir, map_location="cpu", weights_only=False)["models"][0])
    if args.train_guards_only:
        L.load_attacker_ensemble([os.path.join(args.attacker_load_dir, "ep%d.pt" % c) for c in args.attacker_ckpts])
    if rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
    num_updates = args.num_frames // args.num_steps // (E * world)   # train_fortattack.py:197
    L.reset()
    start = time.time()
    shift = args.ckpt + 1 if args.continue_training else 0           # train_fortattack.py:42-43
    for j in range(shift, num_updates + shift):
        L.collect()
        vals = L.update(train_guards_only=args.train_guards_only)
        L.after_update()
        if rank == 0 and j % args.save_interval == 0:
            L.save(os.path.join(args.save_dir, "ep%d.pt" % j))       # train_fortattack.py:121-128
        if rank == 0 and j % args.log_interval == 0:
            total = (j + 1 - shift) * E * world * args.num_steps
            row, n_ep = eng.eval_stats()
            print(json.dumps({"update": j, "num_timesteps": total, "fps": int(total / (time.time() - start)),
                              "value_loss": float(vals[0, 0]), "action_loss": float(vals[0, 1]),
                              "dist_entropy": float(vals[0, 2]),
                              "total_reward_per_agent": L.episode_rewards.mean(0).tolist(),
                              "episodes": n_ep, "guards_win_rate": float(row[2]), "fort_reached_rate": float(row[3])}),
                  flush=True)
    L.close()                                          # communicators / per-team process groups: on every rank, here
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
