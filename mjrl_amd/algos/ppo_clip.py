"""PPO (clipped surrogate) with the minibatch-Adam epochs on the GPU.

Mirror of the reference class (mjrl/algos/ppo_clip.py:22-110): same constructor arguments,
`PPO_surrogate`, `train_from_paths`.  The batch is ingested once (page-locked staging), the
`epochs * (N // mb_size)` minibatch steps -- index draws by `np.random.choice`, the reference's
random stream -- run through `mjx_policy_minibatch_adam` with the old policy fixed; surrogate
and KL before / after come from the same K3 kernel NPG / TRPO use.

Parity note: in the reference the old NETWORK silently follows the new one during the epochs once
`policy.set_param_values` has been called with a float32 array (its new / old tensors then alias
one buffer; only `old_log_std` stays fixed), i.e. from the second training iteration on.  With
`reference_aliasing=True` (default) this class reproduces those numbers; `False` keeps the old
policy fixed, as the algorithm is published.
"""
import time as timer

import numpy as np

from .._lib import check, ptr
from ..utils.ingest import minibatch_indices, upload
from ..utils.logger import DataLog
from .batch_reinforce import BatchREINFORCE


class PPO(BatchREINFORCE):
    def __init__(self, env, policy, baseline, clip_coef=0.2, epochs=10, mb_size=64, learn_rate=3e-4, seed=123,
                 save_logs=False, reference_aliasing=True, **kwargs):
        self.reference_aliasing = reference_aliasing
        self.env = env
        self.policy = policy
        self.baseline = baseline
        self.learn_rate = learn_rate
        self.seed = seed
        self.save_logs = save_logs
        self.clip_coef = clip_coef
        self.epochs = epochs
        self.mb_size = mb_size
        self.running_score = None
        if save_logs:
            self.logger = DataLog()
        self._adam = None           # (m, v, steps): the state of the reference's torch.optim.Adam, kept across iterations

    def PPO_surrogate(self, observations, actions, advantages):
        """mean(min(LR adv, clamp(LR) adv)) of the current new / old parameters -- ppo_clip.py:49-56 (host value, NumPy fp32)"""
        p = self.policy
        obs, act, adv = np.float32(observations), np.float32(actions), np.float32(advantages)

        def LL(model, ls):
            z = (act - model.forward(obs)) / np.exp(ls)
            return -0.5 * np.sum(z ** 2, axis=1) - np.sum(ls) - 0.5 * p.m * np.log(2 * np.pi)
        LR = np.exp(LL(p.model, np.float32(p.get_param_values()[-p.m:])) - LL(p.old_model, np.float32(p.get_old_param_values()[-p.m:])))
        return float(np.mean(np.minimum(LR * adv, np.clip(LR, 1 - self.clip_coef, 1 + self.clip_coef) * adv)))

    def train_from_paths(self, paths):
        """ppo_clip.py:59-110"""
        from ..engine import _dist
        from ..utils import ranks
        base_stats = self._process_and_bind(paths)             # concatenation-free ingestion, whitened advantages, statistics
        if self.save_logs:
            self.log_rollout_statistics(paths)
        eng = self.engine
        torch = eng.torch
        surr_before = eng.eval_surr_kl()[0]
        ts = timer.time()
        # One process per GPU (r05): minibatch Adam is a sequential chain over rows drawn from the WHOLE batch, not a sum over
        # samples -- like the MLP baseline's fit, every rank gathers all ranks' fp32 rows (rank order = the one-process batch) and
        # runs the IDENTICAL chain from the same index draws (the last rank's: it sampled the batch's last episodes, so its NumPy
        # stream stands where a single process's would; every rank draws, the streams advance alike).  The copies stay
        # bit-identical; surrogate / KL before and after are rank sums over the shards as everywhere else.
        obs_b, act_b, adv_b = eng.obs, eng.act, eng.adv
        if _dist() is not None:
            obs_b, act_b, adv_b = (ranks.gather_rows(t[:eng.N_local]).contiguous() for t in (eng.obs, eng.act, eng.adv))
        num_samples = int(obs_b.shape[0])
        steps = self.epochs * int(num_samples / self.mb_size)
        if steps > 0:
            idx = minibatch_indices(eng.lib, num_samples, steps, self.mb_size)
            idx = ranks.broadcast_host(idx, src=-1)
            if self._adam is None:
                self._adam = [torch.zeros_like(eng.theta_new), torch.zeros_like(eng.theta_new), 0]
            didx = upload(eng.backend, idx)
            # reference_aliasing: reproduce what the reference computes once its new / old network tensors share memory
            # (policies/gaussian_mlp.py set_param_values); False = the old policy stays fixed during the epochs
            track = int(bool(self.reference_aliasing and getattr(self.policy, "reference_new_old_alias", False)))
            check(eng.lib.mjx_policy_minibatch_adam(eng.ctx, 2, ptr(obs_b), ptr(act_b), ptr(adv_b), ptr(didx), steps,
                                                    self.mb_size, ptr(eng.theta_new), ptr(eng.tr_new), ptr(eng.theta_old),
                                                    ptr(eng.tr_old), track, ptr(self._adam[0]), ptr(self._adam[1]), self._adam[2],
                                                    self.learn_rate, self.clip_coef, None, eng.stream()))
            self._adam[2] += steps
            eng.old_is_new = False
            eng._bind_policy()
        surr_after, kl_dist = eng.eval_surr_kl()
        self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)
        t_opt = timer.time() - ts
        if self.save_logs:
            self.logger.log_kv('t_opt', t_opt)
            self.logger.log_kv('kl_dist', kl_dist)
            self.logger.log_kv('surr_improvement', surr_after - surr_before)
            self.logger.log_kv('running_score', self.running_score)
            self._log_success(paths)
        self.last_update = dict(kl_dist=kl_dist, surr_before=surr_before, surr_after=surr_after)
        return base_stats

    def __getstate__(self):
        state = super().__getstate__()
        state["_adam"] = None
        return state
