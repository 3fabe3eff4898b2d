"""Demo-augmented policy gradient (DAPG) on the GPU.

Mirrors ``mjrl.algos.dapg.DAPG`` (reference mjrl/algos/dapg.py:25-141): the gradient is the
vanilla gradient over [on-policy ; demonstrations] with demo "advantages"
lam_0 * lam_1^iter, rescaled by N_all / N; the Fisher metric and the surrogate / KL use the
on-policy block only.  Both blocks are uploaded once; K1 runs over all rows, K2/K3 over the
on-policy prefix of the same device arrays.
"""
import time as timer

import numpy as np

from ..utils.logger import DataLog
from .npg_cg import NPG


class DAPG(NPG):
    def __init__(self, env, policy, baseline, demo_paths=None, normalized_step_size=0.01,
                 FIM_invert_args={'iters': 10, 'damping': 1e-4}, hvp_sample_frac=1.0, seed=123, save_logs=False,
                 kl_dist=None, lam_0=1.0, lam_1=0.95, **kwargs):
        """Arguments as in the reference (dapg.py:26-52)."""
        self.env = env
        self.policy = policy
        self.baseline = baseline
        self.kl_dist = kl_dist if kl_dist is not None else 0.5 * normalized_step_size
        self.seed = seed
        self.save_logs = save_logs
        self.FIM_invert_args = FIM_invert_args
        self.hvp_subsample = hvp_sample_frac
        self.running_score = None
        self.demo_paths = demo_paths
        self.lam_0 = lam_0
        self.lam_1 = lam_1
        self.iter_count = 0.0
        self.input_normalization = None
        self.alpha = None
        if save_logs:
            self.logger = DataLog()

    def train_from_paths(self, paths):
        """dapg.py:54-141"""
        # advantages (whitened over all ranks) and path statistics on the host; observations / actions of the on-policy
        # paths and of the demonstrations go path by path through the page-locked stager (no concatenated host copies)
        advantages, base_stats, self.running_score = self._advantages_and_statistics(paths)
        if self.save_logs:
            self.log_rollout_statistics(paths)
        N = advantages.shape[0]
        use_demos = self.demo_paths is not None and self.lam_0 > 0.0
        batch = list(paths)
        if use_demos:                                          # dapg.py:62-70
            from ..engine import _dist
            d = _dist()
            demos = self.demo_paths if d is None else self.demo_paths[d.get_rank()::d.get_world_size()]   # shard demos too
            n_demo = int(sum(len(p["observations"]) for p in demos))
            demo_adv = self.lam_0 * (self.lam_1 ** self.iter_count) * np.ones(n_demo)
            self.iter_count += 1
            batch = batch + list(demos)
            all_adv = 1e-2 * np.concatenate([advantages / (self._global_mean_std(advantages)[1] + 1e-8), demo_adv])
        else:
            all_adv = advantages

        eng = self.engine
        staged = eng.stage_paths(batch, ("observations", "actions"))
        self._push_policy()
        eng.set_batch(staged["observations"], staged["actions"], all_adv)   # [on-policy ; demos] uploaded once
        N_all_global = eng.N_global
        N_on_global = eng.global_count(N)

        subsampled = self.hvp_subsample is not None and self.hvp_subsample < 0.99
        if not subsampled and eng.old_is_new:
            # the whole update in ONE call into libmjx (mjx_dapg_update): K1 over all rows, gradient x N_all / N_on, the
            # on-policy prefix bound with its own advantages, K3, CG, step length, step, K3 -- one read-back
            t0 = timer.time()
            res = eng.dapg_update(self.FIM_invert_args['iters'], self.FIM_invert_args['damping'], 2.0 * self.kl_dist,
                                  self.policy.min_log_std, N, advantages, N_on_global=N_on_global)
            if res is not None:
                surr_after, kl_dist = res
                late = eng.deferred()
                surr_before, gdotx, alpha = late["surr_before"], late["gdotx"], late["alpha"]
                self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)
                if self.save_logs:
                    self._log_update(paths, alpha, 2.0 * self.kl_dist, 0.0, timer.time() - t0, kl_dist, surr_before, surr_after)
                self.last_update = dict(alpha=float(alpha), kl_dist=kl_dist, surr_before=surr_before, surr_after=surr_after,
                                        gdotx=gdotx)
                return base_stats

        t0 = timer.time()
        g, _ = eng.surr_vpg()                                  # K1 over all rows, mean over N_all
        g.mul_(N_all_global / N_on_global)                     # sample_coef, dapg.py:97-98
        t_gLL = timer.time() - t0

        # Fisher / surrogate / KL: on-policy prefix only, means over the on-policy count
        eng.bind_rows(N, adv=advantages, N_global=N_on_global)
        surr_before = eng.eval_surr_kl()[0]                    # dapg.py:92 (theta_new == theta_old here)
        t0 = timer.time()
        _, gdotx = self.CG_solve(g)                            # dapg.py:103-106
        t_FIM = timer.time() - t0

        n_step_size = 2.0 * self.kl_dist                       # dapg.py:111-112
        alpha = np.sqrt(np.abs(n_step_size / (gdotx + 1e-20)))
        eng.apply_step(alpha, self.policy.min_log_std)
        surr_after, kl_dist = eng.eval_surr_kl()
        self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)

        if self.save_logs:
            self._log_update(paths, alpha, n_step_size, t_gLL, t_FIM, kl_dist, surr_before, surr_after)
        self.last_update = dict(alpha=float(alpha), kl_dist=kl_dist, surr_before=surr_before, surr_after=surr_after,
                                gdotx=gdotx)
        return base_stats
