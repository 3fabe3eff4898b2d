"""TRPO = NPG direction + backtracking line search on the KL constraint.

Mirrors ``mjrl.algos.trpo.TRPO`` (reference mjrl/algos/trpo.py:24-146).  Each line-search
trial is one K3 launch (surrogate + KL at theta + alpha * x) and one scalar read-back;
parameters never leave the device until the search ends.
"""
import time as timer

import numpy as np

from ..utils.logger import DataLog
from .npg_cg import NPG


class TRPO(NPG):
    def __init__(self, env, policy, baseline, kl_dist=0.01, FIM_invert_args={'iters': 10, 'damping': 1e-4},
                 hvp_sample_frac=1.0, seed=123, save_logs=False, normalized_step_size=0.01, **kwargs):
        """Arguments as in the reference (trpo.py:25-54)."""
        self.env = env
        self.policy = policy
        self.baseline = baseline
        self.kl_dist = kl_dist if kl_dist is not None else 0.5 * normalized_step_size
        self.seed = seed
        self.save_logs = save_logs
        self.FIM_invert_args = FIM_invert_args
        self.hvp_subsample = hvp_sample_frac
        self.running_score = None
        self.input_normalization = None
        self.alpha = None
        if save_logs:
            self.logger = DataLog()

    def train_from_paths(self, paths):
        """trpo.py:56-146"""
        base_stats = self._process_and_bind(paths)
        if self.save_logs:
            self.log_rollout_statistics(paths)
        eng = self.engine

        subsampled = self.hvp_subsample is not None and self.hvp_subsample < 0.99
        if not subsampled and eng.old_is_new:
            # K1, CG, step length and the line search in libmjx (mjx_trpo_update): trials enqueued in batches, the accept /
            # shrink decision on the device, one read-back per batch, up to the reference's 100 trials (then alpha = 0).
            # (torch.distributed fallback: the loop below.)
            t0 = timer.time()
            iters, damping = self.FIM_invert_args['iters'], self.FIM_invert_args['damping']
            res = eng.trpo_update(iters, damping, 2.0 * self.kl_dist, self.kl_dist, self.policy.min_log_std)
            if res is not None:
                late = eng.deferred()
                surr_before, gdotx = late["surr_before"], late["gdotx"]
                for (sa, klk) in (res["history"][:-1] if res["accepted"] else res["history"]):
                    print("Step size too high. Backtracking. | kl = %f | surr diff = %f" % (klk, sa - surr_before))
                alpha, trials, surr_after, kl_dist = res["alpha"], res["trials"], res["surr_after"], res["kl"]
                self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)
                if self.save_logs:
                    self._log_update(paths, alpha, 2.0 * self.kl_dist, 0.0, timer.time() - t0, kl_dist, surr_before, surr_after)
                self.last_update = dict(alpha=float(alpha), kl_dist=kl_dist, surr_before=surr_before, surr_after=surr_after,
                                        gdotx=gdotx, trials=trials)
                return base_stats

        t0 = timer.time()
        g, surr_before = eng.surr_vpg()
        t_gLL = timer.time() - t0
        t0 = timer.time()
        _, gdotx = self.CG_solve(g)
        t_FIM = timer.time() - t0

        n_step_size = 2.0 * self.kl_dist
        alpha = np.sqrt(np.abs(n_step_size / (gdotx + 1e-20)))

        # backtracking: accept the first alpha with KL < kl_dist, shrink by 0.9, give up (alpha = 0)
        # after 100 trials -- trpo.py:107-120
        trials = 0
        for k in range(100):
            eng.apply_step(alpha, self.policy.min_log_std)
            surr_after, kl_dist = eng.eval_surr_kl()
            trials += 1
            if kl_dist < self.kl_dist:
                break
            alpha = 0.9 * alpha
            print("Step size too high. Backtracking. | kl = %f | surr diff = %f" % (kl_dist, surr_after - surr_before))
            if k == 99:
                alpha = 0.0
        eng.apply_step(alpha, self.policy.min_log_std)        # trpo.py:122-126
        surr_after, kl_dist = eng.eval_surr_kl()
        self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)

        if self.save_logs:
            self._log_update(paths, alpha, n_step_size, t_gLL, t_FIM, kl_dist, surr_before, surr_after)
        self.last_update = dict(alpha=float(alpha), kl_dist=kl_dist, surr_before=surr_before, surr_after=surr_after,
                                gdotx=gdotx, trials=trials)
        return base_stats
