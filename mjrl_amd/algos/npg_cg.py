"""Natural policy gradient with a conjugate-gradient Fisher solve, on the GPU.

Mirrors ``mjrl.algos.npg_cg.NPG`` (reference mjrl/algos/npg_cg.py:23-163): same constructor
keywords, same step-size rule, same logged keys.  ``train_from_paths`` issues

    K1  surrogate + vanilla gradient            (1 launch; the reference does 5 passes)
    K4  CG: iters x (K2 Fisher-vector product + device-side vector update), no host sync
    K3  surrogate + KL at the stepped parameters

through ``UpdateEngine`` and reads back two scalars (g.x, then surr/KL) plus the new
parameter vector.  The north-star name ``CG_solve`` is kept as a public method.
"""
import time as timer

import numpy as np

from ..utils.cg_solve import DeviceFisher
from ..utils.logger import DataLog
from .batch_reinforce import BatchREINFORCE


class NPG(BatchREINFORCE):
    def __init__(self, env, policy, baseline, normalized_step_size=0.01, const_learn_rate=None,
                 FIM_invert_args={'iters': 10, 'damping': 1e-4}, hvp_sample_frac=1.0, seed=123, save_logs=False,
                 kl_dist=None, input_normalization=None, **kwargs):
        """Arguments as in the reference (npg_cg.py:24-60)."""
        self.env = env
        self.policy = policy
        self.baseline = baseline
        self.alpha = const_learn_rate
        self.n_step_size = normalized_step_size if kl_dist is None else 2.0 * kl_dist
        self.seed = seed
        self.save_logs = save_logs
        self.FIM_invert_args = FIM_invert_args
        self.hvp_subsample = hvp_sample_frac
        self.running_score = None
        if save_logs:
            self.logger = DataLog()
        self.input_normalization = input_normalization
        if self.input_normalization is not None and not (0 < self.input_normalization <= 1):
            self.input_normalization = None

    # ------------------------------------------------------------------ Fisher pieces
    def HVP(self, observations, actions, vector, regu_coef=None):
        """(H + regu I) vector on host ndarrays -- npg_cg.py:62-81."""
        regu_coef = self.FIM_invert_args['damping'] if regu_coef is None else regu_coef
        obs = observations
        if self.hvp_subsample is not None and self.hvp_subsample < 0.99:
            idx = np.random.choice(observations.shape[0], size=int(self.hvp_subsample * observations.shape[0]))
            obs = observations[idx]
        self._push_policy()
        self.engine.set_batch(obs)
        return DeviceFisher(self.engine, regu_coef)(vector)

    def build_Hvp_eval(self, inputs, regu_coef=None):
        """npg_cg.py:83-88; the returned operator is recognised by mjrl_amd.utils.cg_solve."""
        regu_coef = self.FIM_invert_args['damping'] if regu_coef is None else regu_coef
        self._push_policy()
        self.engine.set_batch(inputs[0], inputs[1] if len(inputs) > 1 else None)
        return DeviceFisher(self.engine, regu_coef)

    def CG_solve(self, b, iters=None, damping=None, sync=True):
        """x = (H + damping I)^-1 b by CG on the currently bound batch; b is a device tensor or
        a host vector; returns (x device tensor, b.x).  sync=False leaves b.x on the device (engine.deferred())."""
        eng = self.engine
        if not hasattr(b, "data_ptr"):
            b = eng.to_device_f32(np.asarray(b, np.float32))
        iters = self.FIM_invert_args['iters'] if iters is None else iters
        damping = self.FIM_invert_args['damping'] if damping is None else damping
        if self.hvp_subsample is not None and self.hvp_subsample < 0.99:
            return self._cg_subsampled(b, iters, damping)
        return eng.cg_solve(b, iters, damping, sync=sync)

    def _cg_subsampled(self, b, iters, damping):
        """hvp_sample_frac < 0.99: a fresh with-replacement row sample per product, drawn from
        NumPy's global RNG exactly like npg_cg.py:65-69.  Rows are drawn from the rows that are BOUND (DAPG binds the
        on-policy prefix of its [on-policy ; demonstrations] block before the solve, dapg.py:103), and the engine's
        block / prefix binding is restored afterwards."""
        eng, be, torch = self.engine, self.engine.backend, self.engine.torch
        Nb = eng.N_bound
        k = int(self.hvp_subsample * Nb)
        from ..engine import _dist
        from ..utils.ingest import upload
        d = _dist()
        kg = eng.global_count(k)
        be.cg_init(b)
        try:
            for _ in range(int(iters)):
                idx = upload(be, np.random.choice(Nb, size=k))
                sub = eng.obs.index_select(0, idx)
                be.bind_batch(sub, None, None, k, kg)
                be.fvp_of_cg_direction(eng.Ap)
                if d is not None:
                    d.all_reduce(eng.Ap)
                be.cg_step(eng.Ap, damping, 1e-10)
            be.cg_finish(b, eng.x, eng.bdotx)
        finally:
            eng.rebind()                                       # back to the whole shard (and DAPG's on-policy prefix)
        return eng.x, float(eng.bdotx.item())

    # ------------------------------------------------------------------ update
    def _normalize_inputs(self, observations):
        """running input normalisation, npg_cg.py:101-107 (touches only policy.model)."""
        m = self.policy.model
        mean, std = self._global_column_mean_std(observations)
        shift = self.input_normalization * m.in_shift + (1 - self.input_normalization) * mean
        scale = self.input_normalization * m.in_scale + (1 - self.input_normalization) * std
        m.set_transformations(shift, scale, m.out_shift, m.out_scale)

    def _log_update(self, paths, alpha, n_step_size, t_gLL, t_FIM, kl_dist, surr_before, surr_after):
        self.logger.log_kv('alpha', alpha)
        self.logger.log_kv('delta', n_step_size)
        self.logger.log_kv('time_vpg', t_gLL)
        self.logger.log_kv('time_npg', t_FIM)
        self.logger.log_kv('kl_dist', kl_dist)
        self.logger.log_kv('surr_improvement', surr_after - surr_before)
        self.logger.log_kv('running_score', self.running_score)
        self._log_success(paths)

    def train_from_paths(self, paths):
        """npg_cg.py:91-163"""
        if self.input_normalization:
            # the running input normalisation needs the column statistics of the host observations (npg_cg.py:101-107)
            observations, actions, advantages, base_stats, self.running_score = self.process_paths(paths)
            self._normalize_inputs(observations)
            self._bind(observations, actions, advantages)
            stats_later = None
        else:
            stats_later = self._process_and_bind(paths, defer_stats=True)      # (the path statistics: under the update, below)
        eng = self.engine

        const_alpha = self.alpha is not None
        subsampled = self.hvp_subsample is not None and self.hvp_subsample < 0.99
        iters, damping = self.FIM_invert_args['iters'], self.FIM_invert_args['damping']
        if not subsampled and eng.old_is_new:
            # The whole update in ONE call into libmjx (mjx_npg_update): K1, CG, step length formed on the device from g.x,
            # step, K3 -- rank sums included -- and one read-back.  (t_gLL / t_FIM cannot be told apart any more: the
            # gradient time is logged as 0, the solve time is the whole call.)
            t0 = timer.time()
            eng.npg_update(iters, damping, self.n_step_size, self.policy.min_log_std,
                           const_alpha=self.alpha if const_alpha else None, enqueue_only=True)      # npg_cg.py:108-141
            # host work the update does not depend on runs under its device time: per-path return statistics, rollout log entries
            if stats_later is not None:
                base_stats, stats_later = stats_later(), None
            if self.save_logs:
                self.log_rollout_statistics(paths)
            surr_after, kl_dist = eng.npg_update_result()
            t_gLL, t_FIM = 0.0, timer.time() - t0
        else:
            if stats_later is not None:
                base_stats, stats_later = stats_later(), None
            if self.save_logs:
                self.log_rollout_statistics(paths)
            # row-subsampled Fisher products (a fresh host-drawn sample per product, npg_cg.py:65-69) and the general
            # position theta_new != theta_old (input_normalization, :101-107): call by call
            t0 = timer.time()
            g, _ = eng.surr_vpg(sync=False)                   # npg_cg.py:111-115
            t_gLL = timer.time() - t0
            t0 = timer.time()
            self.CG_solve(g, sync=const_alpha or subsampled)  # npg_cg.py:120-123
            t_FIM = timer.time() - t0
            if const_alpha:
                eng.apply_step(self.alpha, self.policy.min_log_std)    # npg_cg.py:137-139
            else:
                eng.apply_npg_step(self.n_step_size, self.policy.min_log_std)   # alpha = sqrt(|delta / (g.x + 1e-20)|), on the device
            surr_after, kl_dist = eng.eval_surr_kl()          # npg_cg.py:140-141
        late = eng.deferred()
        surr_before, gdotx = late["surr_before"], late["gdotx"]
        if const_alpha:                                       # npg_cg.py:128-130
            alpha = self.alpha
            n_step_size = (alpha ** 2) * gdotx
        else:
            alpha, n_step_size = late["alpha"], self.n_step_size
        self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)

        if self.save_logs:
            self._log_update(paths, alpha, n_step_size, t_gLL, t_FIM, kl_dist, surr_before, surr_after)
        self.last_update = dict(alpha=float(alpha), kl_dist=kl_dist, surr_before=surr_before, surr_after=surr_after,
                                gdotx=gdotx)
        return base_stats
