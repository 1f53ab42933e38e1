"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run in the BUILD CONTAINER only (needs /root/reference):   python oracle/gen_golden.py
The reference is imported unmodified through oracle/ref_harness.py; this script only
*drives* it (np.random.seed, env.reset/step, Learner/RolloutStorage/JointPPO calls in
the order of train_fortattack.py:49-116) and records inputs and outputs.  The fixtures
are data (actions in, observations / rewards / flags / storage tensors out); no
reference source text is stored.

Batch "identical seeds" convention: env e of a batch with
base_seed B is the reference run as  np.random.seed(B + e); construct; reset();
step(actions[t, e]); on done reset() (same RNG stream continues).
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def scripted_actions(rng, T, E, G, A, p_shoot=0.35, p_up=0.45):
    """Aggressive scripted-random policy so that all three endings and many deaths occur."""
    N = G + A
    a = rng.randint(0, 8, size=(T, E, N))
    m = rng.rand(T, E, N)
    a[:, :, :G] = np.where(m[:, :, :G] < p_shoot, 7, a[:, :, :G])
    a[:, :, G:] = np.where(m[:, :, G:] < p_up, 3, a[:, :, G:])
    # a few attackers shoot back
    a[:, :, G:] = np.where(m[:, :, G:] > 0.9, 7, a[:, :, G:])
    return a.astype(np.int8)


def run_env_batch(G, A, max_t, T, E, base_seed, act_seed, full):
    """Drive E reference envs sequentially (the numpy RNG is global)."""
    N = G + A
    actions = scripted_actions(np.random.RandomState(act_seed), T, E, G, A)
    rec = dict(actions=actions, obs0=np.zeros((E, N, 6)), reward=np.zeros((T, E, N)),
               done=np.zeros((T, E), np.uint8), alive_before=np.zeros((T, E, N), np.uint8),
               hit=np.zeros((T, E, N), np.uint8), was_hit=np.zeros((T, E, N), np.uint8),
               game_result=np.zeros((T, E, 3), np.uint8), obs_sum=np.zeros((T, E)),
               alive_after=np.zeros((T, E, N), np.uint8))
    if full:
        rec["obs"] = np.zeros((T, E, N, 6))  # trainer view: post-reset obs where done
    term_obs, term_idx = [], []
    final = dict(prev_dist=np.zeros((E, N)), num_hit=np.zeros((E, N), np.int32),
                 num_was_hit=np.zeros((E, N), np.int32), time_step=np.zeros(E, np.int32))
    skip = None
    for e in range(E):
        np.random.seed(base_seed + e)
        env, skip = rh.make_reference_env(G, A, max_t)
        with rh.quiet():
            obs = env.reset()
        rec["obs0"][e] = obs
        for t in range(T):
            rec["alive_before"][t, e] = obs[:, 0]
            with rh.quiet():
                obs, rew, done, _ = env.step(actions[t, e].astype(np.int64))
            ag = env.world.agents
            rec["reward"][t, e] = np.array(rew, dtype=np.float64)
            rec["done"][t, e] = done
            rec["alive_after"][t, e] = obs[:, 0]
            live = [(a.alive or a.justDied) for a in ag]
            rec["hit"][t, e] = [int(a.hit and l) for a, l in zip(ag, live)]
            rec["was_hit"][t, e] = [int(a.wasHit and l) for a, l in zip(ag, live)]
            if done:
                rec["game_result"][t, e] = env.world.gameResult
                term_obs.append(obs.copy())
                term_idx.append((t, e))
                with rh.quiet():
                    obs = env.reset()
            rec["obs_sum"][t, e] = obs.sum()
            if full:
                rec["obs"][t, e] = obs
        ag = env.world.agents
        final["prev_dist"][e] = [np.nan if a.prevDist is None else a.prevDist for a in ag]
        final["num_hit"][e] = [a.numHit for a in ag]
        final["num_was_hit"][e] = [a.numWasHit for a in ag]
        final["time_step"][e] = env.world.time_step
    rec["term_obs"] = np.array(term_obs).reshape(-1, N, 6)
    rec["term_idx"] = np.array(term_idx, np.int32).reshape(-1, 2)
    rec.update({"final_" + k: v for k, v in final.items()})
    rec["meta"] = np.array([G, A, max_t, T, E, base_seed, skip], np.int64)
    return rec


def gen_env_fixtures():
    specs = [  # name, G, A, max_t, T, E, base_seed, act_seed, full
        ("env_3v3", 3, 3, 60, 192, 4, 0, 11, True),
        ("env_5v5", 5, 5, 100, 160, 3, 123, 12, True),
        ("env_2v4", 2, 4, 40, 96, 2, 77, 13, True),
        ("env_1v1", 1, 1, 30, 64, 2, 5, 14, True),
        ("env_3v3_long", 3, 3, 100, 400, 32, 4096, 15, False),
        ("env_5v5_long", 5, 5, 100, 300, 16, 900, 16, False),
    ]
    for name, G, A, max_t, T, E, bs, as_, full in specs:
        rec = run_env_batch(G, A, max_t, T, E, bs, as_, full)
        gr = rec["game_result"].reshape(-1, 3).sum(0)
        print("%-14s steps=%d deaths=%d endings[dead,timeout,fort]=%s" % (
            name, T * E, int((rec["alive_before"] > rec["alive_after"]).sum()), gr.tolist()))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def gen_mt_kat():
    seeds = np.array([0, 1, 123, 4096, 2 ** 31 - 1, 2 ** 32 - 1], np.uint64)
    first = np.zeros((len(seeds), 8))
    after = np.zeros((len(seeds), 8))
    for i, s in enumerate(seeds):
        np.random.seed(int(s))
        first[i] = np.random.random_sample(8)
        np.random.random_sample(700 - 8)  # crosses two state regenerations (624 words = 312 doubles)
        after[i] = np.random.random_sample(8)
    np.random.seed(123)
    uni = np.array([np.random.uniform(-1, 1, 1)[0], np.random.uniform(-0.8, 0.8 * -0.8, 1)[0],
                    np.random.uniform(-0.8 * 0.15 / 2, 0.8 * 0.15 / 2, 1)[0],
                    np.random.uniform(0.8 * 0.8, 0.8, 1)[0]])
    np.savez(os.path.join(OUT, "mt19937_kat.npz"), seeds=seeds, first=first, after_700=after,
             uniform_seed123=uni)
    print("mt19937_kat   ", first[2][:2])


def _args(T, device="cpu"):
    import torch
    ns = argparse.Namespace(
        env_name="fortattack-v1", num_agents=3, mask_dist=None, entity_mp=False, identity_size=0,
        num_processes=1, num_steps=T, num_env_steps=25, no_cuda=True, cuda=False,
        device=torch.device(device), dist_threshold=0.1, arena_size=1, lr=1e-4, gamma=0.99, tau=0.95,
        entropy_coef=0.01, value_loss_coef=0.5, max_grad_norm=0.5, ppo_epoch=1, num_mini_batch=4,
        clip_param=0.2, clipped_value_loss=True, continue_training=False, attacker_load_dir=None,
        attacker_ckpts=[], load_dir=None)
    return ns


def gen_collector_fixture(name="collector_3v3", G=3, A=3, T=64, max_t=25, n_upd=3, seed=2024, compact=False, load_ckpt=None):
    """Drive the reference's own Learner/Neo/RolloutStorage/JointPPO through the call
    sequence of train_fortattack.py:49-116 for a few updates and record everything the
    collector, GAE and advantage normalisation produce.  `compact` (the long-horizon capture:
    the reference's own rollout length, arguments.py:23 / marlsave/tmp_2/params.json): the storage
    after `after_update` is recorded as its row 0 + the absolute sum of the other rows, and the
    int64 action tensor is dropped (`actions` int8 holds the same values).  `load_ckpt`: one of
    the published checkpoints, loaded through the reference's own `Learner.load_models`
    (learner.py:71-73: `continue_training`) so that episodes end at irregular steps (trained
    attackers reach the fort, trained guards shoot them) instead of every max_t steps."""
    import torch
    rh.import_reference()
    torch.set_num_threads(1)
    import learner as ref_learner
    import rlcore.algo.ppo as ref_ppo

    N = G + A
    torch.manual_seed(seed)
    np.random.seed(seed)
    env, skip = rh.make_reference_env(G, A, max_t)
    args = _args(T)
    master = ref_learner.setup_master(args, env)
    if load_ckpt is not None:
        master.load_models(torch.load(load_ckpt, weights_only=False, map_location="cpu")["models"])

    captured = {}
    orig_gen = ref_ppo.magent_feed_forward_generator

    def capturing_gen(rollouts_list, opp_rollouts_list, advantages_list, num_mini_batch):
        captured.setdefault("adv", []).append([a.clone() for a in advantages_list])
        return orig_gen(rollouts_list, opp_rollouts_list, advantages_list, num_mini_batch)

    ref_ppo.magent_feed_forward_generator = capturing_gen

    rec = dict(actions=np.zeros((n_upd, T, N), np.int8), done=np.zeros((n_upd, T), np.uint8))
    keys = ["obs", "rewards", "masks", "value_preds", "returns", "action_log_probs"] + ([] if compact else ["actions_st"])
    for k in keys + ["adv"] + (["after_obs_row0", "after_masks_row0", "after_rest_abs_sum"] if compact else ["after_obs", "after_masks"]):
        rec[k] = []
    end_pts_all, next_values_all = [], []
    with rh.quiet():
        obs = env.reset()
    rec["obs0"] = obs.copy()
    for j in range(n_upd):
        end_pts = []
        master.initialize_obs(obs)
        step = 0
        while step < T:
            masks = torch.FloatTensor(obs[:, 0])
            with torch.no_grad():
                actions_list, _ = master.act(step, masks)
            agent_actions = np.array(actions_list).reshape(-1)
            rec["actions"][j, step] = agent_actions
            with rh.quiet():
                obs, reward, done, _ = env.step(agent_actions)
            reward = torch.from_numpy(np.stack(reward)).float()
            master.update_rollout(obs, reward, masks)
            rec["done"][j, step] = done
            step += 1
            if done:
                end_pts.append(step)
                with rh.quiet():
                    obs = env.reset()
                masks = torch.FloatTensor(obs[:, 0])
                master.initialize_new_episode(step, obs, masks)
        if end_pts[-1] != T:
            end_pts.append(T)
        master.wrap_horizon(end_pts)
        st = [a.rollouts for a in master.all_agents]
        # value_preds[end_pt] after wrap_horizon == the next_value used for that segment
        next_values_all.append(np.array([[float(s.value_preds[ep, 0, 0]) for ep in end_pts] for s in st],
                                        np.float32))
        end_pts_all.append(end_pts)
        for k in keys:
            attr = "actions" if k == "actions_st" else k
            rec[k].append(np.stack([getattr(s, attr).numpy().copy() for s in st]))
        captured.pop("adv", None)
        with rh.quiet():
            master.update()
        adv = [None] * N  # guards trained first, then attackers (learner.py:180-184)
        adv[:G] = [a.numpy() for a in captured["adv"][0]]
        adv[G:] = [a.numpy() for a in captured["adv"][1]]
        rec["adv"].append(np.stack(adv))
        master.after_update()
        if compact:
            assert all(bool(torch.equal(a.actions[:, 0, 0], torch.from_numpy(rec["actions"][j, :, i].astype(np.int64))))
                       for i, a in enumerate(st))
            rec["after_obs_row0"].append(np.stack([s.obs[0].numpy().copy() for s in st]))
            rec["after_masks_row0"].append(np.stack([s.masks[0].numpy().copy() for s in st]))
            # masks[1:] keep the rollout's values (storage.py:51-56 only moves row T to row 0); obs[1:] are zeroed
            rec["after_rest_abs_sum"].append(np.array([float(s.obs[1:].abs().sum()) for s in st]))
        else:
            rec["after_obs"].append(np.stack([s.obs.numpy().copy() for s in st]))
            rec["after_masks"].append(np.stack([s.masks.numpy().copy() for s in st]))
    ref_ppo.magent_feed_forward_generator = orig_gen
    for k in list(rec.keys()):
        if isinstance(rec[k], list):
            rec[k] = np.stack(rec[k])  # (n_upd, N, ...)
    max_seg = max(len(e) for e in end_pts_all)
    ep = np.full((n_upd, max_seg), -1, np.int32)
    nv = np.zeros((n_upd, N, max_seg), np.float32)
    for j, e in enumerate(end_pts_all):
        ep[j, :len(e)] = e
        nv[j, :, :len(e)] = next_values_all[j]
    rec["end_pts"], rec["next_values"] = ep, nv
    rec["meta"] = np.array([G, A, max_t, T, n_upd, seed, skip], np.int64)
    rec["gamma_tau"] = np.array([args.gamma, args.tau])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, " end_pts", end_pts_all)


def gen_mpnn_fixture():
    """Reference MPNN (mpnn.py) at hidden_dim=32: weights + inputs -> value, action
    log-probs under given actions, entropy.  Small enough (~10k params) to ship."""
    import torch
    rh.import_reference()
    torch.set_num_threads(1)
    from mpnn import MPNN

    class _Sp(object):
        shape = (8,)

    torch.manual_seed(7)
    rec = {}
    for tag, n_own, n_opp in (("g", 3, 3), ("a", 5, 2)):
        net = MPNN(action_space=_Sp(), num_agents=n_own, num_opp_agents=n_opp, num_entities=0,
                   input_size=6, hidden_dim=32, pos_index=2, mask_dist=None, entity_mp=False,
                   policy_layers=1)
        B = 5
        inp = torch.randn(n_own * B, 6)      # agent-major (learner.py:150)
        opp = torch.randn(n_opp * B, 6)
        act = torch.randint(0, 8, (n_own * B, 1))
        with torch.no_grad():
            value, logp, ent, _ = net.evaluate_actions(inp, None, opp, None, act)
            x = net._fwd(inp, opp, None)
            logits = net.dist(net._policy(x)).logits
        for k, v in net.state_dict().items():
            rec["%s.sd.%s" % (tag, k)] = v.numpy()
        rec[tag + ".inp"], rec[tag + ".opp"], rec[tag + ".act"] = inp.numpy(), opp.numpy(), act.numpy()
        rec[tag + ".value"], rec[tag + ".logp"] = value.numpy(), logp.numpy()
        rec[tag + ".entropy"], rec[tag + ".logits"] = ent.numpy(), logits.numpy()
        rec[tag + ".shape"] = np.array([n_own, n_opp, B, 32])
    # parameter count of the full-size policy (158 153)
    full = MPNN(action_space=_Sp(), num_agents=3, num_opp_agents=3, num_entities=0, input_size=6,
                pos_index=2, mask_dist=None, entity_mp=False, policy_layers=1)
    rec["full_param_count"] = np.array(sum(p.numel() for p in full.parameters()))
    rec["full_keys"] = np.array(list(full.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "mpnn_h32.npz"), **rec)
    print("mpnn_h32       params(full)=%d" % int(rec["full_param_count"]))


def mpnn_h128_setup(pol, torch):
    """What both sides do to a freshly constructed full-size policy before the h = 128 golden forward (here on the
    reference module, in tests/ on the repo's): non-zero biases and larger logits than the 0.01-gain init gives, drawn
    from torch's global stream in parameter order."""
    for p in pol.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.3, 0.3)
    pol.dist.linear.weight.data.mul_(3.0)


def mpnn_fingerprint(pol, np):
    """(P, 4) float64 per parameter tensor in state_dict order: sum, sum of absolute values, first and last element
    (numpy float64 sums).  Compared with a tolerance: orthogonal_ init goes through a LAPACK QR whose last bits
    depend on the thread count / CPU dispatch, so the seed gives the same weights to ~1e-7, not to the bit."""
    out = []
    for v in pol.state_dict().values():
        a = v.detach().cpu().numpy().reshape(-1).astype(np.float64)
        out.append([a.sum(), np.abs(a).sum(), a[0], a[-1]])
    return np.array(out, np.float64)


def gen_mpnn_h128_fixture():
    """The reference MPNN at its full size (hidden_dim 128, mpnn.py:18-89), both teams, 3v3 and 5v5: seed-constructed
    weights (NOT shipped: 158 153 floats per policy; the repo's module draws the same ones from the same seed -- the
    fingerprints in the fixture prove it before anything is compared) + observations -> value, log-softmax of the
    logits.  Closes the chain reference -> mpnn.py -> fa_policy_kernel at h = 128 on the GPU box."""
    import torch
    rh.import_reference()
    torch.set_num_threads(1)
    from mpnn import MPNN

    class _Sp(object):
        shape = (8,)

    rec = {}
    for tag, G, A, seed, B in (("3v3", 3, 3, 21, 96), ("5v5", 5, 5, 22, 60)):
        N = G + A
        torch.manual_seed(seed)
        nets = []
        for n, m in ((G, A), (A, G)):
            net = MPNN(action_space=_Sp(), num_agents=n, num_opp_agents=m, num_entities=0, input_size=6, pos_index=2,
                       mask_dist=None, entity_mp=False, policy_layers=1)
            mpnn_h128_setup(net, torch)
            nets.append(net)
        g = torch.Generator().manual_seed(seed + 100)
        obs = torch.randn((B, N, 6), generator=g)
        obs[:, :, 0] = (torch.rand((B, N), generator=g) > 0.3).float()
        obs[:, :, 3] = obs[:, :, 3] * 3 + 4.7
        value, logp = torch.zeros(B, N), torch.zeros(B, N, 8)
        with torch.no_grad():
            for net, own, opp in ((nets[0], slice(0, G), slice(G, N)), (nets[1], slice(G, N), slice(0, G))):
                n = own.stop - own.start
                inp = obs[:, own].transpose(0, 1).reshape(-1, 6)      # agent-major, as learner.py:150-152 cats
                oin = obs[:, opp].transpose(0, 1).reshape(-1, 6)
                x = net._fwd(inp, oin, None)
                v = net._value(x)
                lg = net.dist(net._policy(x)).logits                  # normalised logits = log-softmax
                value[:, own] = v.view(n, B).transpose(0, 1)
                logp[:, own] = lg.view(n, B, 8).transpose(0, 1)
        rec[tag + ".meta"] = np.array([G, A, seed, B], np.int64)
        rec[tag + ".obs"] = obs.numpy()
        rec[tag + ".value"], rec[tag + ".logp_all"] = value.numpy(), logp.numpy()
        rec[tag + ".fingerprint_g"] = mpnn_fingerprint(nets[0], np)
        rec[tag + ".fingerprint_a"] = mpnn_fingerprint(nets[1], np)
    np.savez_compressed(os.path.join(OUT, "mpnn_h128.npz"), **rec)
    print("mpnn_h128      ", {k: v.shape for k, v in rec.items() if k.endswith("obs")})


def tensor_fingerprint(a, np):
    """[sum, sum of absolute values, first, last, l2 norm] of a tensor in float64."""
    a = a.detach().cpu().numpy().reshape(-1).astype(np.float64)
    return [a.sum(), np.abs(a).sum(), a[0], a[-1], np.sqrt((a * a).sum())]


def gen_ppo_update_fixture():
    """The reference's JointPPO.update (rlcore/algo/ppo.py:116-204) on seed-constructed FULL-SIZE policies (h = 128;
    weights not shipped -- the fingerprint scheme of mpnn_h128.npz proves the repo's module draws the same ones), ONE
    full-batch minibatch (ppo_epoch 1, num_mini_batch 1: the sampler's permutation only reorders a sum), 3v3 and 5v5,
    both teams, clipped and un-clipped value loss.  Recorded: the rollout rows in the joint (B, N, .) layout, the
    normalised advantages (ppo.py:121-123), the three losses update() returns, and -- read off the reference's
    parameters after the call -- the clipped gradients (p.grad after clip_grad_norm_) and the Adam displacement
    (weights after - before): fingerprints of every tensor, every element of the small ones, a strided sample of the
    large ones.  Closes the chain reference -> fa_ppo_grad / fa_adam_step on the GPU box without a hop through this
    repo's autograd restatement."""
    import torch
    rh.import_reference()
    torch.set_num_threads(1)
    from mpnn import MPNN
    from rlcore.algo.ppo import JointPPO
    from rlcore.storage import RolloutStorage

    class _Sp(object):
        shape = (8,)

    clip, vcoef, ecoef, lr = 0.2, 0.5, 0.01, 1e-4                  # arguments.py:22-45 defaults
    rec = {}
    # (max_grad_norm 0.5 is the default; the gradient norms here are 0.2 .. 0.5, so two cases lower it to make
    #  clip_grad_norm_ actually scale and one raises it so that it does not)
    # (small batches on purpose: see the relu-kink note in run_case -- 48 / 50 own rows keep the number of relu inputs low
    #  enough that a seed without a near-zero pre-activation exists)
    cases = [("3v3_g_clip", 3, 3, 0, True, 31, 2, 8, 5.0), ("3v3_a_noclip", 3, 3, 1, False, 32, 2, 8, 0.1),
             ("5v5_g_clip", 5, 5, 0, True, 33, 2, 5, 0.5), ("5v5_a_clip", 5, 5, 1, True, 34, 2, 5, 0.15)]
    import copy

    def run_case(G, A, team, clipped, seed, T, P, gnorm):
        N = G + A
        n, m = (G, A) if team == 0 else (A, G)
        own = slice(0, G) if team == 0 else slice(G, N)
        opp = slice(G, N) if team == 0 else slice(0, G)
        torch.manual_seed(seed)
        net = MPNN(action_space=_Sp(), num_agents=n, num_opp_agents=m, num_entities=0, input_size=6, pos_index=2,
                   mask_dist=None, entity_mp=False, policy_layers=1)
        mpnn_h128_setup(net, torch)
        fingerprint = mpnn_fingerprint(net, np)
        g = torch.Generator().manual_seed(seed + 100)
        B = T * P
        obs = torch.randn((T + 1, P, N, 6), generator=g)
        obs[..., 0] = (torch.rand((T + 1, P, N), generator=g) > 0.3).float()      # alive flags = the loss masks (ppo.py:224)
        obs[..., 3] = obs[..., 3] * 3 + 4.7
        actions = torch.randint(0, 8, (T, P, N, 1), generator=g)
        value_preds = torch.randn((T + 1, P, N, 1), generator=g)
        returns = value_preds + 0.6 * torch.randn((T + 1, P, N, 1), generator=g)   # some inside, some outside the value clip
        # old log-probs: the policy's own log-probs of these actions + noise, so that the ratios straddle [1-clip, 1+clip]
        with torch.no_grad():
            flat = lambda t, sl: t[:T, :, sl].reshape(B, -1, t.shape[-1]).transpose(0, 1).reshape(-1, t.shape[-1])
            _, lp, _, _ = net.evaluate_actions(flat(obs, own), None, flat(obs, opp), None, flat(actions, own))
        old_logp = torch.randn((T, P, N, 1), generator=g)
        old_logp[:, :, own] = lp.view(n, T, P, 1).permute(1, 2, 0, 3) + 0.25 * torch.randn((T, P, n, 1), generator=g)
        adv = torch.zeros((T, P, N, 1))
        for i in range(N):                                           # ppo.py:121-123, per agent
            a = returns[:-1, :, i] - value_preds[:-1, :, i]
            adv[:, :, i] = (a - a.mean()) / (a.std() + 1e-5)

        def storage(i, dtype):
            s = RolloutStorage(T, P, (6,), None, 1)
            for k in ("obs", "recurrent_hidden_states", "rewards", "value_preds", "returns", "action_log_probs", "masks"):
                setattr(s, k, getattr(s, k).to(dtype))
            s.obs.copy_(obs[:, :, i])
            s.actions.copy_(actions[:, :, i])
            s.action_log_probs.copy_(old_logp[:, :, i])
            s.value_preds.copy_(value_preds[:, :, i])
            s.returns.copy_(returns[:, :, i])
            return s

        def update(model, dtype, threads):
            """the reference's JointPPO.update on a copy of the policy -> (losses, clipped gradients, Adam displacement)"""
            torch.set_num_threads(threads)
            pol = copy.deepcopy(model).to(dtype)
            before = [p.detach().clone() for p in pol.parameters()]
            ppo = JointPPO(pol, clip, 1, 1, vcoef, ecoef, lr=lr, max_grad_norm=gnorm, use_clipped_value_loss=clipped)
            with rh.quiet():
                losses = ppo.update([storage(i, dtype) for i in range(own.start, own.stop)],
                                    [storage(i, dtype) for i in range(opp.start, opp.stop)])
            torch.set_num_threads(1)
            grads = [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in pol.parameters()]
            return losses, grads, [p.detach() - b for p, b in zip(pol.parameters(), before)], pol

        losses, grads, deltas, pol = update(net, torch.float32, 1)
        # A relu whose pre-activation is within float32 rounding of 0 for some sample passes or blocks that sample's gradient
        # depending on the summation order of the implementation -- both outcomes are "right", and they differ by a whole
        # single-sample term (~1e-2 of a small tensor's largest entry; seen: the first 5v5 fixture agreed with the reference on
        # one CPU and not on another, nor on the GPU).  A fixture must not sit on such a kink: (a) in float64 no relu input of
        # the forward is closer to 0 than 1e-5 (float32 rounding of a pre-activation is ~1e-6); (b) the same update in
        # float64 and with another GEMM blocking (8 threads) gives the float32 gradients back.
        margin = [float("inf")]
        net64 = copy.deepcopy(net).double()
        hooks = [mod.register_forward_pre_hook(lambda _m, inp: margin.__setitem__(0, min(margin[0], float(inp[0].abs().min()))))
                 for mod in net64.modules() if isinstance(mod, torch.nn.ReLU)]
        with torch.no_grad():
            net64.evaluate_actions(flat(obs, own).double(), None, flat(obs, opp).double(), None, flat(actions, own))
        for h in hooks:
            h.remove()
        worst = 0.0
        for other in (update(net, torch.float64, 1)[1], update(net, torch.float32, 8)[1]):
            for a_, b_ in zip(grads, other):
                if float(a_.abs().max()) > 0:
                    worst = max(worst, float((a_.double() - b_.double()).abs().max() / a_.abs().max()))
        return dict(N=N, n=n, own=own, B=B, fingerprint=fingerprint, obs=obs, actions=actions, value_preds=value_preds, returns=returns,
                    old_logp=old_logp, adv=adv, losses=losses, grads=grads, deltas=deltas, names=[k for k, _ in net.named_parameters()],
                    lp=lp, kink=worst, margin=margin[0])

    for tag, G, A, team, clipped, seed, T, P, gnorm in cases:
        while True:
            r = run_case(G, A, team, clipped, seed, T, P, gnorm)
            if r["kink"] < 1e-4 and r["margin"] > 1e-5:
                break
            print("ppo_update %-13s seed %d sits on a relu kink (smallest |relu input| %.1e; float64 / 8-thread gradients differ by %.1e): "
                  "next seed" % (tag, seed, r["margin"], r["kink"]))
            seed += 1000
        N, n, own, B = r["N"], r["n"], r["own"], r["B"]
        obs, actions, value_preds, returns, old_logp, adv, lp = [r[k] for k in ("obs", "actions", "value_preds", "returns", "old_logp", "adv", "lp")]
        vl, al, ent = r["losses"]
        rec[tag + ".fingerprint"] = r["fingerprint"]
        rec[tag + ".meta"] = np.array([G, A, team, int(clipped), seed, T, P], np.int64)
        rec[tag + ".relu_margin"] = np.array(r["margin"])
        rec[tag + ".hyper"] = np.array([clip, vcoef, ecoef, lr, gnorm], np.float64)
        rec[tag + ".obs"] = obs[:T].reshape(B, N, 6).numpy()
        rec[tag + ".actions"] = actions.reshape(B, N, 1).numpy().astype(np.int8)
        rec[tag + ".value_preds"] = value_preds[:T].reshape(B, N, 1).numpy()
        rec[tag + ".returns"] = returns[:T].reshape(B, N, 1).numpy()
        rec[tag + ".old_logp"] = old_logp.reshape(B, N, 1).numpy()
        rec[tag + ".adv"] = adv.reshape(B, N, 1).numpy()
        rec[tag + ".losses"] = np.array([vl, al, ent], np.float64)
        names = r["names"]
        rec[tag + ".param_names"] = np.array(names)
        gfp, dfp = [], []
        for k, grad, delta in zip(names, r["grads"], r["deltas"]):
            # (a parameter the forward never touches -- oppUpdate, mpnn.py:44 -- has no gradient and does not move)
            gfp.append(tensor_fingerprint(grad, np))
            dfp.append(tensor_fingerprint(delta, np))
            stride = 1 if grad.numel() <= 1024 else 61               # small tensors whole, large ones sampled
            rec["%s.grad.%s" % (tag, k)] = grad.reshape(-1)[::stride].numpy()
            rec["%s.delta.%s" % (tag, k)] = delta.reshape(-1)[::stride].numpy()
        rec[tag + ".grad_fingerprint"] = np.array(gfp, np.float64)
        rec[tag + ".delta_fingerprint"] = np.array(dfp, np.float64)
        ratio_out = float(((torch.exp(lp.view(n, T, P, 1).permute(1, 2, 0, 3) - old_logp[:, :, own]) - 1).abs() > clip).float().mean())
        print("ppo_update %-13s seed %d losses=%s |g|=%.4f ratios outside the clip range: %.0f%%; float64 / 8-thread gradients within %.1e, smallest |relu input| %.1e" % (
            tag, seed, np.round(rec[tag + ".losses"], 5).tolist(), float(np.sqrt((np.array(gfp)[:, 4] ** 2).sum())), 100 * ratio_out, r["kink"], r["margin"]))
    np.savez_compressed(os.path.join(OUT, "ppo_update_h128.npz"), **rec)


def gen_attackers_fixture():
    """BASELINE config 5 with the PUBLISHED policies: the attacker state_dicts (entry -1 of 'models', learner.py:137-139) of
    the reference's shipped checkpoints marlsave/tmp_1/ep{220,650,1240,1600,2520}.pt (arguments.py --attacker-ckpts) --
    weights are data -- together with the reference mpnn.py's outputs for them on a fixed 5v5 observation batch."""
    import torch
    rh.import_reference()
    torch.set_num_threads(1)
    from mpnn import MPNN

    class _Sp(object):
        shape = (8,)

    eps = [220, 650, 1240, 1600, 2520]
    G = A = 5
    N, B = G + A, 64
    g = torch.Generator().manual_seed(505)
    obs = torch.randn((B, N, 6), generator=g)
    obs[:, :, 0] = (torch.rand((B, N), generator=g) > 0.25).float()
    obs[:, :, 1:3] = obs[:, :, 1:3].clamp(-1, 1) * 0.8
    obs[:, :, 3] = obs[:, :, 3] * 2 + 3.1
    rec = {"episodes": np.array(eps, np.int64), "obs": obs.numpy(), "meta": np.array([G, A, B], np.int64)}
    for e in eps:
        ck = torch.load(os.path.join(rh.REFERENCE_ROOT, "marlsave", "tmp_1", "ep%d.pt" % e), map_location="cpu", weights_only=False)
        sd = ck["models"][-1]
        net = MPNN(action_space=_Sp(), num_agents=A, num_opp_agents=G, num_entities=0, input_size=6, pos_index=2,
                   mask_dist=None, entity_mp=False, policy_layers=1)
        net.load_state_dict(sd)
        for k, v in sd.items():
            rec["ep%d.%s" % (e, k)] = v.numpy()
        with torch.no_grad():
            inp = obs[:, G:].transpose(0, 1).reshape(-1, 6)          # agent-major (learner.py:150-152)
            oin = obs[:, :G].transpose(0, 1).reshape(-1, 6)
            x = net._fwd(inp, oin, None)
            rec["ep%d.out.value" % e] = net._value(x).view(A, B).transpose(0, 1).numpy()
            rec["ep%d.out.logp_all" % e] = net.dist(net._policy(x)).logits.view(A, B, 8).transpose(0, 1).numpy()
    np.savez_compressed(os.path.join(OUT, "attackers_tmp1.npz"), **rec)
    print("attackers_tmp1 ", eps, "max p(action) = %.3f" % max(float(np.exp(rec["ep%d.out.logp_all" % e]).max()) for e in eps))


def gen_choice_fixture():
    """The ensemble path's RNG interleaving: train_fortattack_v2.py follows EVERY
    env.reset() -- the first one (:29-35) and each episode-end one (:104-111) -- with
    master.sample_attacker(), whose only RNG use is `np.random.choice(self.attacker_ckpts)`
    (learner.py:120) on the same global stream the resets draw from.  Driven here exactly so: reference env,
    reset, np.random.choice(ckpts), scripted steps, on done reset + choice.  Records the checkpoint index
    chosen after every reset and the observation after every step (post-reset where done)."""
    G, A, max_t, T, E, base_seed, K = 5, 5, 25, 120, 4, 700, 5
    ckpts = [220, 650, 1240, 1600, 2520]                 # arguments.py default --attacker-ckpts
    N = G + A
    actions = scripted_actions(np.random.RandomState(21), T, E, G, A)
    rec = dict(actions=actions, obs0=np.zeros((E, N, 6)), obs=np.zeros((T, E, N, 6)), done=np.zeros((T, E), np.uint8),
               choice0=np.zeros(E, np.int32), choice=np.full((T, E), -1, np.int32), reward=np.zeros((T, E, N)))
    skip = None
    for e in range(E):
        np.random.seed(base_seed + e)
        env, skip = rh.make_reference_env(G, A, max_t)
        with rh.quiet():
            obs = env.reset()
        rec["obs0"][e] = obs
        rec["choice0"][e] = ckpts.index(np.random.choice(ckpts))
        for t in range(T):
            with rh.quiet():
                obs, rew, done, _ = env.step(actions[t, e].astype(np.int64))
            rec["reward"][t, e] = np.array(rew, dtype=np.float64)
            rec["done"][t, e] = done
            if done:
                with rh.quiet():
                    obs = env.reset()
                rec["choice"][t, e] = ckpts.index(np.random.choice(ckpts))
            rec["obs"][t, e] = obs
    rec["meta"] = np.array([G, A, max_t, T, E, base_seed, skip, K], np.int64)
    print("env_choice_5v5 episodes=%d choices=%s" % (int(rec["done"].sum()), np.bincount(rec["choice"][rec["choice"] >= 0], minlength=K).tolist()))
    np.savez_compressed(os.path.join(OUT, "env_choice_5v5.npz"), **rec)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["mt", "env", "collector", "mpnn", "mpnn128", "choice", "ppo", "attackers"]
    if "ppo" in which:
        gen_ppo_update_fixture()
    if "attackers" in which:
        gen_attackers_fixture()
    if "choice" in which:
        gen_choice_fixture()
    if "mt" in which:
        gen_mt_kat()
    if "env" in which:
        gen_env_fixtures()
    if "collector" in which:
        gen_collector_fixture()
    if "collector1000" in which:
        # the reference's own rollout length and native team size (arguments.py:23; marlsave/tmp_2/params.json)
        gen_collector_fixture("collector_5v5_T1000", G=5, A=5, T=1000, max_t=100, n_upd=2, seed=77, compact=True,
                              load_ckpt="/root/reference/marlsave/tmp_1/ep1240.pt")
    if "mpnn" in which:
        gen_mpnn_fixture()
    if "mpnn128" in which:
        gen_mpnn_h128_fixture()
    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.endswith(".npz")}
    print(sizes, "total", sum(sizes.values()))
