"""Batched counterpart of the reference's host loop: Learner / Neo / JointPPO over E envs.

What the reference does per env-step in Python -- cat the per-agent observations
(learner.py:150-152), one MPNN forward per team (:160), chunk the outputs back per agent
(:164-170, with a ``.cpu().numpy()`` per agent), ``env.step``, seven ``copy_`` per agent into
its RolloutStorage (storage.py:33-43) -- is here two launches per env-step: ``fa_collect_act``
(the fused MPNN forward of both teams + sampling, csrc/fa_policy.hip: reads the ``obs[s]`` row, writes
``value_preds[s] / actions[s] / action_log_probs[s]``) and ``fa_collect_step`` (reads ``actions[s]``,
writes ``obs[s+1] / rewards[s] / masks[s+1] / done[s]``).  Nothing leaves the GPU during a rollout and
the whole T-step rollout + V(obs[T]) replays from ONE hipGraph.  ``policy_backend="torch"`` keeps the
forwards in the PyTorch modules (any hidden size; also what an attacker ensemble uses).

  BatchedLearner.collect()      train_fortattack.py:49-110 (rollout + wrap_horizon)
  BatchedLearner.update()       learner.py:175-188 -> JointPPO.update (ppo.py:116-204)
  BatchedLearner.after_update() learner.py:234-236 -> storage.py:51-56
Multi-GPU: env shards per rank; the advantage statistics are all-gathered and merged (dist.py) and
joint_ppo_update() all-reduces one flat buffer per optimizer step (gradients of the un-normalised
losses + the rank's alive-mask mean), which makes every rank take the reference's step on the union
minibatch.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import dist as fa_dist
from .dist import gae_adv_mean_std
from . import mpnn_pack
from .mpnn import MPNN, TwinMPNN
from .storage import JointRolloutStorage


def ppo_losses(pol, own, opp, actions, value_preds, returns, old_logp, adv, clip_param, clipped_value_loss=True,
               normalize=True):
    """The three alive-masked losses of JointPPO.update for one minibatch (ppo.py:146-187).
    own (B,n,6), opp (B,m,6); the rest (B,n,1).  Works on any device, no host sync: the
    reference's `if mask.mean() != 0: x /= mask.mean()` becomes a division by
    where(mean != 0, mean, 1) (x is 0 whenever the mean is 0).
    normalize=False leaves that division out and returns mask.mean() as a fourth value: the
    multi-rank update divides by the mean over ALL ranks' samples after the gradient all-reduce."""
    mask = own[:, :, 0:1]                                      # alive flag = obs[:,0] (ppo.py:224)
    values, logp, ent = pol.evaluate_actions(own, opp, actions)
    mm = mask.mean()
    denom = torch.where(mm != 0, mm, torch.ones_like(mm)) if normalize else torch.ones_like(mm)
    dist_entropy = (ent.unsqueeze(-1) * mask).mean() / denom
    ratio = mask * torch.exp(logp - old_logp)
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param) * adv
    action_loss = (mask * -torch.min(surr1, surr2)).mean() / denom
    if clipped_value_loss:
        vclip = value_preds + (values - value_preds).clamp(-clip_param, clip_param)
        vl = 0.5 * torch.max((values - returns).pow(2), (vclip - returns).pow(2))
    else:
        # ppo.py:178-182: 0.5 * F.mse_loss is already a SCALAR over all samples (dead agents included);
        # multiplying it by the mask, taking the mean and dividing by mask.mean() gives it back
        vl = 0.5 * (returns - values).pow(2).mean()
    value_loss = (vl * mask).mean() / denom
    if not normalize:
        return value_loss, action_loss, dist_entropy, mm
    return value_loss, action_loss, dist_entropy


def _capture_mode():
    """cudaStreamCaptureMode for the hipGraph captures: with a process group alive another thread of the process -- the
    NCCL / RCCL watchdog polling its events -- may call into the runtime while a capture is open; "global" mode turns
    such a call into a capture error, "thread_local" only polices the capturing thread."""
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class GraphedPPOStep(object):
    """One optimizer step of joint_ppo_update -- minibatch forward, backward, (gradient exchange,) clip, Adam --
    captured as hipGraph(s) and replayed with a new minibatch index set.

    `fused` (csrc/fa_train.hip, fa_fold.hip): fold the parameters into weight packs, fa_ppo_grad on rows
    idx of the rollout read in place (forward + losses + backward in one launch), unfold the gradients, fa_adam_step
    on the flat parameter / gradient / moment buffers of mpnn_pack.FlatPolicy: ~16 graph nodes, no autograd.
    `share_cu`: the build of the kernel that leaves room on its CUs (two teams updated concurrently).
    Otherwise PyTorch autograd on a minibatch gathered into static buffers (index_select, outside the graph): ~300
    small kernels whose Python / dispatcher time (4 ms) exceeded their GPU time (2.7 ms at 16 384 x 3 samples)
    before they were replayed from a graph.

    Parameters, gradients and Adam state are the live tensors (the optimizer must be `capturable`).  One rank: one
    graph.  Several ranks: two graphs around the eager all-reduce of the flat buffer (see joint_ppo_update).
    Capture needs warm-up iterations; they run on the first mb rows and are undone (parameters and optimizer state
    restored) before the first real step."""

    def __init__(self, pol, opt, own_sl, opp_sl, rows, mb, clip_param, value_loss_coef, entropy_coef, max_grad_norm,
                 clipped_value_loss, group, fused=False, share_cu=False, exchange=None, adv_stats=None):
        # adv_stats (fused only): persistent (mean, std) float64 (N,) device tensors -> the kernel normalises the advantages
        # itself from returns - value_preds (ppo.py:121-124) and rows[5] is never read
        # the gradient exchange: `exchange` (dist.LibraryExchange: RCCL inside the library) or torch.distributed on
        # `group`; a world of one rank exchanges nothing unless dist.FORCE_COLLECTIVE / an explicit exchange says so
        self.pol, self.opt, self.group = pol, opt, group
        self.world = exchange.world if exchange is not None else _world(group)
        self.exchanging = exchange is not None or fa_dist.exchanging(group)
        self._reduce = exchange.all_reduce_ if exchange is not None else (lambda t: dist.all_reduce(t, group=group))
        self.params = [p for p in pol.parameters()]
        self.fused = bool(fused)
        self.max_grad_norm = max_grad_norm
        # torch path: the minibatch is gathered into static buffers; fused: the kernel reads rows idx of the rollout
        self.static = None if self.fused else [torch.empty((mb,) + tuple(r.shape[1:]), dtype=r.dtype, device=r.device) for r in rows]
        self.rows = tuple(rows)
        self.mb = mb
        world, exchanging = self.world, self.exchanging
        if self.fused:   # csrc/fa_train.hip: forward + losses + backward of the minibatch in one launch
            from .env import ppo_grad
            dev = rows[0].device
            N = rows[0].shape[1]
            n_own = own_sl.stop - own_sl.start
            team, G = (0, n_own) if own_sl.start == 0 else (1, N - n_own)
            self.fp = mpnn_pack.FlatPolicy.of(pol)
            self.fp.bind_adam(opt)
            self.idx = torch.zeros(mb, dtype=torch.int64, device=dev)
            self._out = torch.zeros(mpnn_pack.SLAB_FLOATS, device=dev)
            self._scratch = None
            inv_count = 1.0 / (mb * n_own)
            PF, LOSS = mpnn_pack.PF_FLOATS, mpnn_pack.PLAIN_FLOATS
            self._unmask = torch.tensor([0.0, 1.0, 1.0], device=dev)

        def fused_fwd_bwd():
            """parameters -> weight packs (fold: 3 launches) -> fa_ppo_grad on rows idx of the rollout (mask mean,
            forward + losses + backward, reduction: 3) -> gradients of the parameters (unfold: 2): no PyTorch
            autograd, GEMM or gather; left to torch are four scalar ops on the loss sums"""
            fp = self.fp
            w, wt = fp.fold_pack()
            _, self._scratch = ppo_grad(*self.rows[:5], None if adv_stats is not None else self.rows[5], w, wt, None, team, G, N - G,
                                        clip_param, value_loss_coef, entropy_coef, clipped_value_loss, scratch=self._scratch,
                                        out=self._out, idx=self.idx, normalize=not exchanging, share_cu=share_cu, adv_stats=adv_stats)
            fp.attach_grads()               # every parameter's .grad is its slice of fp.gflat
            fp.unfold(self._out)
            sums = self._out[LOSS:LOSS + 3] * inv_count
            mmp = self._out[LOSS + 9]       # the alive-mask mean of the minibatch (1 where that is 0)
            if not clipped_value_loss:      # the scalar-MSE value loss is not masked (ppo.py:178-182)
                sums = sums * (self._unmask + (1.0 - self._unmask) * mmp)
            if not exchanging:
                return sums / mmp, None
            fp.gflat[PF:PF + 3].copy_(sums)  # the flat gradient buffer carries the loss sums and the mask mean
            fp.gflat[PF + 3].copy_(self._out[LOSS + 3] * inv_count)
            return sums, fp.gflat

        def fwd_bwd():
            if self.fused:
                return fused_fwd_bwd()
            obs_b, act_b, vp_b, ret_b, olp_b, adv_b = self.static
            out = ppo_losses(pol, obs_b[:, own_sl], obs_b[:, opp_sl], act_b[:, own_sl], vp_b[:, own_sl], ret_b[:, own_sl],
                             olp_b[:, own_sl], adv_b[:, own_sl], clip_param, clipped_value_loss, normalize=not exchanging)
            opt.zero_grad(set_to_none=True)
            (out[0] * value_loss_coef + out[1] - out[2] * entropy_coef).backward()
            losses = torch.stack([out[0].detach(), out[1].detach(), out[2].detach()])
            if not exchanging:
                return losses, None
            grads = [p.grad for p in self.params if p.grad is not None]
            return losses, torch.cat([g.reshape(-1) for g in grads] + [losses, out[3].detach().reshape(1)])

        def finish(losses, flat):
            if flat is not None and self.fused:               # in place: flat IS the gradients (fp.gflat)
                flat.div_(world)
                mm = flat[PF + 3].clone()
                flat.div_(torch.where(mm != 0, mm, torch.ones_like(mm)))
                losses = flat[PF:PF + 3]
            elif flat is not None:                            # after the all-reduce (sum over ranks)
                flat = flat / world
                mm = flat[-1]
                flat = flat / torch.where(mm != 0, mm, torch.ones_like(mm))
                losses = flat[-4:-1]
                off = 0
                for p in self.params:
                    if p.grad is not None:
                        p.grad.copy_(flat[off:off + p.grad.numel()].view_as(p.grad))
                        off += p.grad.numel()
            if self.fused:      # parameters, gradients and Adam state are flat buffers: clip + Adam = 2 launches
                self.fp.adam_step(opt, max_grad_norm)
            else:
                nn.utils.clip_grad_norm_(self.params, max_grad_norm)
                opt.step()
            return losses.clone()

        def eager():
            losses, flat = fwd_bwd()
            if flat is not None:
                self._reduce(flat)
            return finish(losses, flat)

        # ---- warm-up on a side stream, then undo it -------------------------------------------------------
        if self.fused:
            self.idx.copy_(torch.arange(mb, device=rows[0].device))
        else:
            for st, src in zip(self.static, rows):
                st.copy_(src[:mb])
        saved_p = [p.detach().clone() for p in self.params]
        had_state = len(opt.state) > 0
        saved_s = {p: {k: v.clone() for k, v in opt.state[p].items() if torch.is_tensor(v)} for p in self.params if p in opt.state}
        side = torch.cuda.Stream(rows[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, q in zip(self.params, saved_p):
                p.copy_(q)
            for p in self.params:
                for k, v in opt.state.get(p, {}).items():
                    if torch.is_tensor(v):
                        v.copy_(saved_s[p][k]) if had_state and p in saved_s else v.zero_()
        # ---- capture -----------------------------------------------------------------------------------
        self.g1 = torch.cuda.CUDAGraph()
        self.g2 = None
        if not exchanging:
            with torch.cuda.graph(self.g1, capture_error_mode=_capture_mode()):
                self.losses = eager()
        else:
            with torch.cuda.graph(self.g1, capture_error_mode=_capture_mode()):
                l0, self.flat = fwd_bwd()
            self.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g2, pool=self.g1.pool(), capture_error_mode=_capture_mode()):
                self.losses = finish(l0, self.flat)

    def run(self, rows, idx, reduce_after=None, reduce_done=None):
        """reduce_after / reduce_done: events that order this step's collective against another chain's on the same
        device (waited for before, recorded after the all-reduce; only used with several ranks)."""
        if self.fused:
            assert all(a.data_ptr() == b.data_ptr() for a, b in zip(rows, self.rows)), "the captured step reads the rollout in place"
            self.idx.copy_(idx)
            self.fp.refresh_hyper(self.opt, self.max_grad_norm)   # lr / betas / eps live in device memory, not in the graph
        else:
            for st, src in zip(self.static, rows):
                torch.index_select(src, 0, idx, out=st)
        self.g1.replay()
        if self.g2 is not None:
            if reduce_after is not None:
                torch.cuda.current_stream().wait_event(reduce_after)
            self._reduce(self.flat)          # between the two graphs, on the same stream
            if reduce_done is not None:
                reduce_done.record()
            self.g2.replay()
        return self.losses


def joint_ppo_update(pol, opt, own_sl, opp_sl, rows, clip_param, ppo_epoch, num_mini_batch, value_loss_coef,
                     entropy_coef, max_grad_norm, clipped_value_loss=True, group=None, sampler=None, graphs=None,
                     exchange=None, adv_stats=None):
    """JointPPO.update (ppo.py:116-204) for one team's shared policy over flattened rollout rows.

    rows = (obs, actions, value_preds, returns, old_log_probs, advantages), each (B, N, .) with B = T * E
    local samples.  A minibatch is one index set applied to every agent of the team and to its opponents
    (magent_feed_forward_generator, ppo.py:207-246).  `sampler(epoch)` -> list of index tensors; default =
    BatchSampler(SubsetRandomSampler(range(B)), int(B / num_mini_batch), drop_last=False) drawn on the device.

    Several ranks (data parallel over env shards): every rank draws minibatches from ITS samples -- the
    global minibatch is their union -- and one flat all-reduce per optimizer step carries the gradients
    of the un-normalised losses plus this rank's alive-mask mean; dividing by the all-rank mask mean
    afterwards gives exactly the gradient of the reference's loss on the union minibatch (the three
    losses are linear in 1 / mask.mean()), so all ranks step identically.
    `exchange`: a dist.LibraryExchange -> the all-reduce is the library's own RCCL call (fa_grad_allreduce) instead of
    torch.distributed's.
    `graphs`: a dict owned by the caller -> full-size minibatch steps replay from hipGraphs (GraphedPPOStep;
    CUDA tensors and a capturable optimizer); a ragged last minibatch runs eagerly.
    Returns a (3,) tensor: (value_loss, action_loss, entropy) summed over the minibatches and divided by
    ppo_epoch * num_mini_batch as the reference does (ppo.py:196-200)."""
    obs_f, act_f, vp_f, ret_f, olp_f, adv_f = rows
    batch = obs_f.shape[0]
    assert batch >= num_mini_batch, (
        "PPO requires the number of processes * number of steps = {} to be greater than "
        "or equal to the number of PPO mini batches ({}).".format(batch, num_mini_batch))
    mb = int(batch / num_mini_batch)                             # ppo.py:210
    world = exchange.world if exchange is not None else _world(group)
    exchanging = exchange is not None or fa_dist.exchanging(group)
    params = [p for p in pol.parameters()]
    acc = torch.zeros(3, device=obs_f.device)
    for epoch in range(ppo_epoch):
        if sampler is not None:
            batches = sampler(epoch)
        else:
            perm = torch.randperm(batch, device=obs_f.device)   # SubsetRandomSampler (ppo.py:213)
            batches = [perm[k:k + mb] for k in range(0, batch, mb)]   # BatchSampler, drop_last=False
        for idx in batches:
            if graphs is not None and idx.numel() == mb:
                key = (id(pol), mb)
                if key not in graphs:
                    graphs[key] = GraphedPPOStep(pol, opt, own_sl, opp_sl, rows, mb, clip_param, value_loss_coef,
                                                 entropy_coef, max_grad_norm, clipped_value_loss, group,
                                                 fused=graphs.get("fused", False) and mpnn_pack.supported(pol),
                                                 exchange=exchange, adv_stats=adv_stats)
                acc += graphs[key].run(rows, idx)
                continue
            obs_b = obs_f[idx]
            out = ppo_losses(pol, obs_b[:, own_sl], obs_b[:, opp_sl], act_f[idx][:, own_sl], vp_f[idx][:, own_sl],
                             ret_f[idx][:, own_sl], olp_f[idx][:, own_sl], adv_f[idx][:, own_sl], clip_param,
                             clipped_value_loss, normalize=not exchanging)
            value_loss, action_loss, dist_entropy = out[:3]
            opt.zero_grad(set_to_none=True)
            (value_loss * value_loss_coef + action_loss - dist_entropy * entropy_coef).backward()
            losses = torch.stack([value_loss.detach(), action_loss.detach(), dist_entropy.detach()])
            if exchanging:
                grads = [p.grad for p in params if p.grad is not None]
                flat = torch.cat([g.reshape(-1) for g in grads] + [losses, out[3].detach().reshape(1)])
                if exchange is not None:
                    exchange.all_reduce_(flat)
                else:
                    dist.all_reduce(flat, group=group)
                flat.div_(world)                                 # equal shard sizes: mean over ranks == union mean
                mm = flat[-1]
                flat.div_(torch.where(mm != 0, mm, torch.ones_like(mm)))
                losses = flat[-4:-1]
                off = 0
                for g in grads:
                    g.copy_(flat[off:off + g.numel()].view_as(g))
                    off += g.numel()
            nn.utils.clip_grad_norm_(params, max_grad_norm)
            opt.step()
            acc += losses
    return acc / (ppo_epoch * num_mini_batch)                    # ppo.py:196-200 (not the batches actually drawn)


class BatchedLearner(object):
    def __init__(self, eng, num_steps=128, hidden_dim=128, lr=1e-4, clip_param=0.2, ppo_epoch=4,
                 num_mini_batch=32, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5,
                 gamma=0.99, tau=0.95, clipped_value_loss=True, use_graph=False, group=None, policy_backend="auto",
                 sample_seed=None, update_backend="auto", exchange="torch", reference_sampling=False):
        # defaults: arguments.py:22-45
        # policy_backend: "hip" = the fused fa_policy kernel runs the rollout's forwards (hidden_dim 128),
        # "torch" = the PyTorch modules do, "auto" = hip whenever it supports the configuration
        self.eng, self.T, self.G, self.A, self.N, self.E = eng, num_steps, eng.G, eng.A, eng.N, eng.E
        self.device = eng.device
        self.gamma, self.tau = gamma, tau
        self.clip_param, self.ppo_epoch, self.num_mini_batch = clip_param, ppo_epoch, num_mini_batch
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.max_grad_norm, self.clipped_value_loss = max_grad_norm, clipped_value_loss
        self.group = group
        # reference_sampling: the minibatch index sets are drawn as the reference draws them -- one torch.randperm(T * E) on the
        # CPU's default generator per epoch and team (BatchSampler(SubsetRandomSampler(range(batch)), mb), ppo.py:213), the
        # guards' epochs first, then the attackers' (learner.py:175-188) -- so that torch.manual_seed(s) gives the reference's
        # minibatches.  Default: permutations drawn on the device (same distribution, no host round trip).  (The ACTION
        # sampling stream cannot be the reference's: FixedCategorical.sample is torch.multinomial on the CPU generator, the
        # engine samples inside the policy kernel from Philox keyed by (seed; rollout, step, env, agent).)
        self.reference_sampling = bool(reference_sampling)
        # exchange: "torch" = torch.distributed collectives on `group` (backend "nccl" is RCCL); "rccl" = the library's own
        # RCCL communicators (dist.LibraryExchange -> fa_adv_allreduce / fa_grad_allreduce), also with ONE rank.
        # Each team's update chain gets its own communicator / process group: the two chains run concurrently on two
        # streams, and collectives of one communicator are ordered
        if exchange not in ("torch", "rccl"):
            raise ValueError("exchange must be 'torch' or 'rccl'")
        self._exch, self._team_exch, self._team_groups = None, [None, None], [group, group]
        if exchange == "rccl":
            self._exch = fa_dist.LibraryExchange(self.device, group)
            self._team_exch = [fa_dist.LibraryExchange(self.device, group) for _ in range(2)]
        elif fa_dist.exchanging(group):
            # dist.new_group is collective over the WHOLE default world: every rank of the job must construct its
            # learner (in the same order), also the ranks outside `group`
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self._team_groups = [dist.new_group(ranks=ranks) for _ in range(2)]
        # learner.py:60-68: guards first (policy1), then attackers (policy2); shared per team
        self.policies = [MPNN(num_agents=self.G, num_opp_agents=self.A, hidden_dim=hidden_dim, num_actions=8),
                         MPNN(num_agents=self.A, num_opp_agents=self.G, hidden_dim=hidden_dim, num_actions=8)]
        for p in self.policies:
            p.to(self.device)
        # ppo.py:114 (capturable: the step count lives on the device, so that an optimizer step can be part of a
        # hipGraph; same arithmetic)
        self.optimizers = [torch.optim.Adam(p.parameters(), lr=lr, capturable=self.device.type == "cuda")
                           for p in self.policies]
        # update_backend "fused" (the default where it applies: use_graph, hidden_dim 128): every optimizer step's
        # forward + losses + backward is the fa_ppo_grad kernel; "torch": PyTorch autograd with the fa_attend op
        if update_backend not in ("auto", "fused", "torch"):
            raise ValueError("update_backend must be 'auto', 'fused' or 'torch'")
        if update_backend == "fused" and not (use_graph and self.device.type == "cuda" and
                                               all(mpnn_pack.supported(p) for p in self.policies)):
            raise ValueError("update_backend='fused' needs use_graph=True, a CUDA device, hidden_dim = 128 and teams of <= 8 "
                             "agents (update_backend='auto' falls back to PyTorch autograd instead)")
        self._update_graphs = {"fused": update_backend != "torch"} if use_graph else None
        self._update_backend = update_backend
        self.storage = JointRolloutStorage(num_steps, self.E, self.N, device=self.device)
        eng.bind_storage(self.storage)
        self.team_slices = [slice(0, self.G), slice(self.G, self.N)]
        self.adv = torch.empty((num_steps, self.E, self.N, 1), device=self.device)
        # the advantage mean / std of the last collect() in buffers that captured update graphs may point at
        self._adv_mean = torch.zeros(self.N, dtype=torch.float64, device=self.device)
        self._adv_std = torch.ones(self.N, dtype=torch.float64, device=self.device)
        self.use_graph = use_graph
        self._graphs = None
        self.episode_rewards = torch.zeros((self.E, self.N), device=self.device)
        self.attacker_pool, self.attacker_id = [], None
        self._twin_net = None
        hip_ok = all(mpnn_pack.supported(p) for p in self.policies)
        if policy_backend not in ("auto", "hip", "torch"):
            raise ValueError("policy_backend must be 'auto', 'hip' or 'torch'")
        if policy_backend == "hip" and not hip_ok:
            raise ValueError("the fused policy kernel needs hidden_dim = 128 and teams of <= 8 agents")
        self.policy_backend = "hip" if (policy_backend != "torch" and hip_ok) else "torch"
        # packed weights of the two policies (rewritten in place whenever the parameters may have moved)
        # the parameters of each policy as one flat buffer + fold / unfold task lists (mpnn_pack.FlatPolicy): the
        # rollout's weight packs and the fused update's gradients come from / go to it without a PyTorch op
        self._flat = [mpnn_pack.FlatPolicy.of(p) for p in self.policies] if (hip_ok and self.device.type == "cuda") else None
        self._packed = [fp.w for fp in self._flat] if (self.policy_backend == "hip" and self._flat) else None
        if self._flat:
            for fp in self._flat:
                fp.fold_pack()
        self.sample_seed = int(torch.initial_seed() if sample_seed is None else sample_seed) & ((1 << 63) - 1)
        self._rollout_counter = torch.zeros(1, dtype=torch.int64, device=self.device)

    def close(self):
        """Release what the constructor took from the communication layer: the library's RCCL communicators
        (ncclCommDestroy) and the per-team process groups.  Idempotent.  Call it (or use the learner as a context manager)
        on every rank at the same point of the program: garbage collection does NOT do it (__del__ only warns)."""
        for ex in [self._exch] + list(self._team_exch):
            if ex is not None:
                ex.close()
        self._exch, self._team_exch = None, [None, None]
        if dist.is_available() and dist.is_initialized():
            for tg in self._team_groups:
                if tg is not None and tg is not self.group:
                    try:
                        dist.destroy_process_group(tg)
                    except Exception:
                        pass
        self._team_groups = [self.group, self.group]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        # collection can happen at any point (another learner's stream capture open, one rank only): nothing here may
        # synchronise or tear down a process group.  close() -- or the context manager -- does that; a learner that
        # still holds communicators when it is collected says so.
        try:
            if self._exch is not None or any(ex is not None for ex in self._team_exch):
                import warnings
                warnings.warn("BatchedLearner collected without close(): its RCCL communicators / per-team process groups "
                              "are left to the process exit", ResourceWarning)
        except Exception:
            pass

    # ---- model I/O (train_fortattack.py:123-128, learner.py:245-249) ----------------------
    def state_dicts(self):
        """[N state_dicts]: entries 0..G-1 the guard policy, G..N-1 the attacker policy."""
        return [self.policies[0].state_dict()] * self.G + [self.policies[1].state_dict()] * self.A

    def load_models(self, policies_list):
        """learner.py:245-249 (load_models): entry 0 -> the guards' policy, entry -1 -> the attackers'.  BOTH state_dicts are
        checked (keys and shapes) before either is loaded: a checkpoint that does not fit leaves the learner as it was."""
        for pol, sd in ((self.policies[0], policies_list[0]), (self.policies[1], policies_list[-1])):
            own = pol.state_dict()
            missing, extra = sorted(set(own) - set(sd)), sorted(set(sd) - set(own))
            if missing or extra:
                raise RuntimeError("load_models: state_dict keys do not match (missing %s, unexpected %s)" % (missing[:3], extra[:3]))
            for k, v in own.items():
                if tuple(sd[k].shape) != tuple(v.shape):
                    raise RuntimeError("load_models: %s has shape %s, the policy expects %s" % (k, tuple(sd[k].shape), tuple(v.shape)))
        self.policies[0].load_state_dict(policies_list[0])
        self.policies[1].load_state_dict(policies_list[-1])

    def save(self, path):
        # the reference's two keys (train_fortattack.py:123-128) + the fused policy kernel's sampling position: the
        # Philox key is (sample_seed; rollout counter, step, env, agent), so a resumed run must not start over at 0
        torch.save({"models": self.state_dicts(), "ob_rms": (None, None),
                    "fa_rollout_counter": int(self._rollout_counter.item()), "fa_sample_seed": int(self.sample_seed)}, path)

    def load(self, path):
        ck = torch.load(path, map_location=self.device, weights_only=False)
        # everything that can refuse the checkpoint is checked BEFORE any state changes: a failed load leaves the learner as it was
        seed = int(ck["fa_sample_seed"]) if "fa_sample_seed" in ck else None
        if seed is not None and seed != self.sample_seed and self._graphs is not None:
            raise RuntimeError("the rollout graph is already captured with another sample_seed: load() before reset()")
        self.load_models(ck["models"])
        if "fa_rollout_counter" in ck:          # (absent in the reference's own checkpoints)
            self._rollout_counter.fill_(int(ck["fa_rollout_counter"]))
        if seed is not None:                    # the sampling stream continues where the saved run stopped
            if seed != self.sample_seed:
                import warnings
                warnings.warn("checkpoint was written with sample_seed %d, this learner was built with %d: continuing the "
                              "checkpoint's sampling stream" % (seed, self.sample_seed))
            self.sample_seed = seed

    # ---- ensemble of frozen attacker strategies (train_fortattack_v2.py, learner.py:119-140) ----
    def load_attacker_ensemble(self, checkpoints, hidden_dim=128):
        """checkpoints: list of paths (reference `ep*.pt` files), checkpoint dicts, or attacker
        state_dicts.  Every env plays against one of the K strategies; the strategy of an env is re-drawn
        each time that env is reset -- the reference calls sample_attacker() after its first env.reset()
        and after every episode-end reset (train_fortattack_v2.py:29-35,104-111), and its
        np.random.choice(attacker_ckpts) (learner.py:120) consumes the same numpy stream the reset positions
        come from (quirk Q14).  The engine reproduces exactly that per env (fa_set_reset_choice): `attacker_id`
        is written by the reset itself.  Train with update(train_guards_only=True).
        With the fused policy kernel the envs are grouped by strategy and each strategy's network runs once
        over its own envs; the PyTorch fallback (policy_backend="torch") evaluates every strategy on every
        env and selects."""
        pool = []
        for ck in checkpoints:
            if isinstance(ck, str):
                ck = torch.load(ck, map_location="cpu", weights_only=False)
            sd = ck["models"][-1] if isinstance(ck, dict) and "models" in ck else ck   # learner.py:137-139
            pol = MPNN(num_agents=self.A, num_opp_agents=self.G, hidden_dim=hidden_dim, num_actions=8)
            pol.load_state_dict(sd)
            pol.to(self.device).eval()
            for p in pol.parameters():
                p.requires_grad_(False)
            pool.append(pol)
        if self._graphs is not None:
            raise RuntimeError("load the ensemble before the first reset() when use_graph is set")
        self.attacker_pool = pool
        self.attacker_id = self.eng.set_reset_choice(len(pool))            # (E,) int32, written by every reset
        self.attacker_id_rows = torch.zeros((self.T, self.E), dtype=torch.int32, device=self.device)
        self._packed_pool = None
        if self.policy_backend == "hip" and all(mpnn_pack.supported(p) for p in pool):
            self._packed_pool = torch.stack([mpnn_pack.pack_policy(p) for p in pool]).contiguous()
        elif self.policy_backend == "hip":
            self.policy_backend = "torch"

    def _attacker_forward(self, fn_name, own, opp):
        """PyTorch fallback: run every strategy on the whole batch and keep, per env, the output of that env's."""
        outs = [getattr(pol, fn_name)(own, opp) for pol in self.attacker_pool]
        sel = self.attacker_id.long()
        pick = lambda ts: torch.stack(ts, 0)[sel, torch.arange(self.E, device=self.device)]
        if isinstance(outs[0], tuple):
            return tuple(pick([o[k] for o in outs]) for k in range(len(outs[0])))
        return pick(outs)

    # ---- rollout -------------------------------------------------------------------------
    def reset(self):
        """env.reset() + initialize_obs (train_fortattack.py:29,49).  With use_graph the rollout's
        hipGraph is captured here first: capture needs a few real warm-up steps, and before the first
        reset the world holds nothing worth keeping."""
        if self.use_graph and self._graphs is None:
            if self.eng.max_time_steps < 8:
                raise ValueError("use_graph needs max_time_steps >= 8 (warm-up steps must not end an episode)")
            self._graphs = self._capture()
        self.eng.collect_reset()

    def _twin(self):
        """One batched pass for both teams when they have the same size (and no ensemble)."""
        if self.G != self.A or self.attacker_pool:
            return None
        if self._twin_net is None:
            self._twin_net = TwinMPNN(self.policies[0], self.policies[1])
        return self._twin_net

    def _hip(self):
        return self.policy_backend == "hip"

    def _hip_act(self, s, value_only=False):
        """One launch: both teams' forward + sampling -> value_preds / actions / action_log_probs[s] (with an
        attacker ensemble: one more small launch that sorts the envs into tiles of equal strategy)."""
        pool = self._packed_pool if self.attacker_pool else None
        self.eng.collect_act(s, self._packed[0], None if pool is not None else self._packed[1], self.sample_seed,
                             self._rollout_counter, value_only=value_only, pool=pool,
                             env_strategy=self.attacker_id if pool is not None else None)

    @torch.no_grad()
    def _act_into_storage(self, s):
        st = self.storage
        if self._hip():
            self._hip_act(s)
            return
        obs = st.obs[s]
        twin = self._twin()
        if twin is not None:
            value, action, logp = twin.act(obs)
            st.value_preds[s].copy_(value)
            st.actions[s].copy_(action)
            st.action_log_probs[s].copy_(logp)
            return
        for pol, own_sl, opp_sl in ((self.policies[0], self.team_slices[0], self.team_slices[1]),
                                    (self.policies[1], self.team_slices[1], self.team_slices[0])):
            if pol is self.policies[1] and self.attacker_pool:
                value, action, logp = self._attacker_forward("act", obs[:, own_sl], obs[:, opp_sl])
            else:
                value, action, logp = pol.act(obs[:, own_sl], obs[:, opp_sl])
            st.value_preds[s, :, own_sl] = value
            st.actions[s, :, own_sl] = action
            st.action_log_probs[s, :, own_sl] = logp

    def step(self, s):
        """One env-step of the rollout: act (learner.py:143-172) + env.step + insert."""
        self._act_into_storage(s)
        if self.attacker_pool:
            self.attacker_id_rows[s].copy_(self.attacker_id)   # which strategy each env faced at step s
        # (an env that ends here is reset by this launch, which also draws its next strategy:
        #  sample_attacker() after every episode-end reset, train_fortattack_v2.py:104-111)
        self.eng.collect_step(s, auto_reset=True)

    def _warm_state(self):
        """A harmless world for the capture warm-up steps: teams on opposite walls facing
        away from each other (no laser can hit, nobody is near the door, no timeout), so no
        episode ends and the reset RNG stream is not touched."""
        import numpy as np
        E, N, G = self.E, self.N, self.G
        x = np.concatenate([np.linspace(-0.9, 0.9, G), np.linspace(-0.9, 0.9, self.A)])
        y = np.concatenate([np.full(G, 0.6), np.full(self.A, -0.6)])
        ang = np.concatenate([np.full(G, np.pi / 2), np.full(self.A, 3 * np.pi / 2)])
        tile = lambda v: np.tile(v[None, :], (E, 1))
        self.eng.set_state(dict(pos_x=tile(x), pos_y=tile(y), vel_x=np.zeros((E, N)), vel_y=np.zeros((E, N)),
                                ang=tile(ang), prev_dist=np.full((E, N), np.nan),
                                alive=np.ones((E, N), np.uint8), time_step=np.zeros(E, np.int32)))
        obs = np.stack([np.ones((E, N)), tile(x), tile(y), tile(ang), np.zeros((E, N)), np.zeros((E, N))], -1)
        self.storage.obs[:2] = torch.from_numpy(obs.astype(np.float32)).to(self.device)

    @torch.no_grad()
    def _value_last(self):
        """wrap_horizon: V(obs[T]) -> value_preds[T] (learner.py:196-202)."""
        st = self.storage
        if self._hip():
            self._hip_act(self.T, value_only=True)
            return
        obs = st.obs[self.T]
        if self._twin() is not None:
            st.value_preds[self.T].copy_(self._twin_net.get_value(obs))
            return
        for pol, own_sl, opp_sl in ((self.policies[0], self.team_slices[0], self.team_slices[1]),
                                    (self.policies[1], self.team_slices[1], self.team_slices[0])):
            if pol is self.policies[1] and self.attacker_pool:
                st.value_preds[self.T, :, own_sl] = self._attacker_forward("get_value", obs[:, own_sl], obs[:, opp_sl])
            else:
                st.value_preds[self.T, :, own_sl] = pol.get_value(obs[:, own_sl], obs[:, opp_sl])

    def _capture(self):
        """ONE hipGraph for the whole rollout: T x (act + env step) and V(obs[T]) -- every launch argument
        is a fixed pointer into the rollout buffers, so the T steps are T different node sets."""
        import numpy as np
        self._warm_state()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for s in (0, 1, 0):  # warm up allocator / rocBLAS handles / code objects on a side stream
                self.step(s)
            self._value_last()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
            for s in range(self.T):
                self.step(s)
            self._value_last()
        # the warm-up steps set prevDist; it survives resets in the reference (quirk Q1), so put
        # back "None" -- everything else is rewritten by the reset that follows
        self.eng.set_state(dict(prev_dist=np.full((self.E, self.N), np.nan)))
        return g

    def refresh_policy_weights(self):
        """Bring every derived copy of the parameters up to date IN PLACE (captured graphs read them):
        the fused kernel's packed buffers, the torch path's fused / stacked operands."""
        if self._flat:
            for fp in self._flat:
                if not fp.attached():   # e.g. a .to() / .data assignment after construction: the kernels would see stale weights
                    raise RuntimeError("a policy's parameters no longer live in its flat buffer (mpnn_pack.FlatPolicy); "
                                       "load weights in place (load_state_dict / copy_) instead of re-pointing them")
                fp.fold_pack()
        for pol in self.policies + list(self.attacker_pool):
            pol.refresh_fused_weights()
        if self._twin() is not None:
            self._twin_net.refresh()

    def collect(self):
        """A T-step rollout from the current obs[0] + GAE (train_fortattack.py:51-110).
        Call reset() before the first rollout and after_update() between rollouts."""
        st = self.storage
        if self.use_graph and self._graphs is None:
            raise RuntimeError("call reset() before collect()")
        self.refresh_policy_weights()
        self._rollout_counter.add_(1)              # a fresh sampling stream for this rollout (fused kernel)
        if self._graphs is not None:
            self._graphs.replay()
        else:
            for s in range(self.T):
                self.step(s)
            self._value_last()
        # compute_returns + the advantage mean / std of ppo.py:121-123 (over ALL ranks) in one pass
        self._adv_mean_std = gae_adv_mean_std(self.eng, self.gamma, self.tau, self.group, exchange=self._exch)
        self._adv_mean.copy_(self._adv_mean_std[0])
        self._adv_std.copy_(self._adv_mean_std[1])
        # train_fortattack.py:88: episode_rewards += reward * masks (alive before the step)
        self.episode_rewards = (st.rewards * st.masks[1:]).sum(0)[..., 0]

    # ---- PPO update (ppo.py:116-204) ------------------------------------------------------------
    def update(self, train_guards_only=False, sampler=None):
        """-> float tensor (n_trained_teams, 3) = mean (value_loss, action_loss, entropy)."""
        st, T, E = self.storage, self.T, self.E
        graphs_with_sampler = False
        if sampler is None and self.reference_sampling:
            sampler, graphs_with_sampler = self._reference_sampler(), True
        mean, std = self._adv_mean_std                           # ppo.py:121-123, from collect()
        # The fused optimizer steps normalise the advantages inside the kernel from (mean, std): the (T, E, N) advantage
        # tensor is only materialised for the paths that read it (PyTorch autograd, a caller's sampler, a ragged last
        # minibatch that runs eagerly)
        batch = T * E
        mb = int(batch / self.num_mini_batch) if batch >= self.num_mini_batch else 0
        g = self._update_graphs
        in_kernel = (g is not None and g.get("fused", False) and (sampler is None or graphs_with_sampler) and mb > 0
                     and batch % mb == 0 and all(mpnn_pack.supported(p) for p in self.policies))
        adv_stats = (self._adv_mean, self._adv_std) if in_kernel else None
        if not in_kernel:
            self.eng.adv_normalize(mean, std, out=self.adv)      # ppo.py:123
        flat = lambda t: t.view(T * E, *t.shape[2:])
        rows = (flat(st.obs[:-1]), flat(st.actions), flat(st.value_preds[:-1]), flat(st.returns[:-1]),
                flat(st.action_log_probs), flat(self.adv))
        out = []
        teams = [0] if train_guards_only else [0, 1]             # learner.py:177
        if len(teams) == 2 and sampler is None and self._teams_step_ok(rows):
            return self._update_teams_together(rows, adv_stats)
        for ti in teams:
            out.append(joint_ppo_update(
                self.policies[ti], self.optimizers[ti], self.team_slices[ti], self.team_slices[1 - ti], rows,
                self.clip_param, self.ppo_epoch, self.num_mini_batch, self.value_loss_coef, self.entropy_coef,
                self.max_grad_norm, self.clipped_value_loss, self._team_groups[ti], sampler,
                graphs=self._update_graphs if (sampler is None or graphs_with_sampler) else None,
                exchange=self._team_exch[ti], adv_stats=adv_stats))
        return torch.stack(out)

    def _reference_sampler(self):
        """magent_feed_forward_generator's index sets (ppo.py:207-213): BatchSampler(SubsetRandomSampler(range(batch)), mb,
        drop_last=False) iterates ONE torch.randperm(batch) of the CPU's default generator in chunks of mb."""
        batch = self.T * self.E
        mb = int(batch / self.num_mini_batch)
        dev = self.device

        def sampler(epoch):
            perm = torch.randperm(batch)
            return [perm[k:k + mb].to(dev) for k in range(0, batch, mb)]
        return sampler

    def _teams_step_ok(self, rows):
        g = self._update_graphs
        batch = rows[0].shape[0]
        mb = int(batch / self.num_mini_batch) if batch >= self.num_mini_batch else 0
        return (g is not None and g.get("fused", False) and g.get("teams_together", True)
                and mb > 0 and batch % mb == 0 and all(mpnn_pack.supported(p) for p in self.policies))

    def _update_teams_together(self, rows, adv_stats=None):
        """Both teams' JointPPO.update (learner.py:175-188 runs them one after the other; they share nothing but the
        read-only rollout) as two concurrent chains of optimizer steps.  The minibatch permutations are drawn in the
        sequential order -- all of the guards' epochs, then all of the attackers' -- so the result is the
        sequential update's, step for step.  Several ranks: each chain's all-reduce (between the two graphs of its step)
        goes through the team's own communicator, so the chains stay independent across the ranks too."""
        g, batch, dev = self._update_graphs, rows[0].shape[0], rows[0].device
        mb = int(batch / self.num_mini_batch)
        perms = [[torch.randperm(batch, device=dev) for _ in range(self.ppo_epoch)] for _ in range(2)]
        steps = []
        for ti in range(2):
            key = (id(self.policies[ti]), mb, "shared")     # (the build of fa_train_kernel that leaves room on its CUs)
            if key not in g:
                g[key] = GraphedPPOStep(self.policies[ti], self.optimizers[ti], self.team_slices[ti], self.team_slices[1 - ti],
                                        rows, mb, self.clip_param, self.value_loss_coef, self.entropy_coef, self.max_grad_norm,
                                        self.clipped_value_loss, self._team_groups[ti], fused=True, share_cu=g.get("share_cu", True),
                                        exchange=self._team_exch[ti], adv_stats=adv_stats)
            steps.append(g[key])
            assert all(a.data_ptr() == b.data_ptr() for a, b in zip(rows, g[key].rows)), "the captured step reads the rollout in place"
        acc = torch.zeros(2, 3, device=dev)
        # Two free-running chains, one stream per team: a team's step is its own graph, nothing joins the teams
        # until the update is over.  While one chain is in the serial part of a step (reduction, unfold, clip,
        # Adam, fold: ~0.2 ms in which it cannot use the GPU) or in the last round of its tiles, the other's
        # tiles fill the CUs.
        # Several ranks: the chains' all-reduces are the only thing they must not do concurrently -- collectives of two
        # communicators in flight on one device can deadlock when the ranks' stream schedulers order them differently --
        # so events chain them in ONE order on every rank (guards' step k, attackers' step k, guards' step k + 1, ...);
        # the graphs on either side of a collective stay concurrent.
        if "team_streams" not in g:
            g["team_streams"] = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        main = torch.cuda.current_stream()
        for st in g["team_streams"]:
            st.wait_stream(main)
        ordered = steps[0].g2 is not None
        last_done = None
        for epoch in range(self.ppo_epoch):
            for k in range(0, batch, mb):
                for ti, st in enumerate(g["team_streams"]):      # (enqueued alternately: both chains start at once)
                    with torch.cuda.stream(st):
                        done = torch.cuda.Event() if ordered else None
                        acc[ti] += steps[ti].run(rows, perms[ti][epoch][k:k + mb], reduce_after=last_done, reduce_done=done)
                        last_done = done
        for st in g["team_streams"]:
            main.wait_stream(st)
        return acc / (self.ppo_epoch * self.num_mini_batch)

    def after_update(self):
        self.eng.after_update()
