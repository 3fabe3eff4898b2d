"""Batch policy-gradient agent base class with mjrl's Agent surface.

Mirrors ``mjrl.algos.batch_reinforce.BatchREINFORCE`` (reference
mjrl/algos/batch_reinforce.py:21-214): ``train_step`` pipeline (sample -> returns ->
advantages -> train_from_paths -> baseline.fit), ``process_paths`` (concatenate + advantage
whitening + return statistics), the surrogate / KL / vanilla-gradient operators and the
vanilla-PG ``train_from_paths``.  All batch arithmetic runs in libmjx through
``mjrl_amd.engine.UpdateEngine``; this class only orchestrates.
"""
import os
import time as timer

import numpy as np

from .. import samplers as trajectory_sampler
from ..engine import UpdateEngine, _dist
from ..utils import process_samples
from ..utils.logger import DataLog


class BatchREINFORCE:
    def __init__(self, env, policy, baseline, learn_rate=0.01, seed=123, desired_kl=None, save_logs=False, **kwargs):
        self.env = env
        self.policy = policy
        self.baseline = baseline
        self.alpha = learn_rate
        self.seed = seed
        self.save_logs = save_logs
        self.running_score = None
        self.desired_kl = desired_kl
        if save_logs:
            self.logger = DataLog()

    # ------------------------------------------------------------------ device plumbing
    _engine_obj = None

    @property
    def engine(self):
        """Lazily created in the training process only (never pickled, never forked)."""
        if self._engine_obj is None:
            self._engine_obj = UpdateEngine(self.policy.n, self.policy.m, self.policy.hidden_sizes)
        return self._engine_obj

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_engine_obj", None)
        state.pop("_stage_pool", None)
        return state

    def _push_policy(self):
        p = self.policy
        self.engine.set_policy(p.get_param_values(), p.get_old_param_values(), p.model.packed_transforms(),
                               p.old_model.packed_transforms())

    def _bind(self, observations, actions, advantages):
        self._push_policy()
        self.engine.set_batch(observations, actions, advantages)

    # ------------------------------------------------------------------ operators (host ndarrays in; same return types as the reference)
    def _scalar(self, v):
        """a 0-dim CPU tensor, like the reference's torch.mean(...) results (its callers unwrap them with
        ``.data.numpy().ravel()[0]``, batch_reinforce.py:139, npg_cg.py:111); float(x) works as well.  fp64: the device
        sums are fp64, nothing is rounded on the way out."""
        return self.engine.torch.tensor(float(v), dtype=self.engine.torch.float64)

    def CPI_surrogate(self, observations, actions, advantages):
        """mean(LR * adv) -- batch_reinforce.py:40-46."""
        self._bind(observations, actions, advantages)
        return self._scalar(self.engine.eval_surr_kl()[0])

    def kl_old_new(self, observations, actions):
        """mean KL(new || old) -- batch_reinforce.py:48-52."""
        self._bind(observations, actions, np.zeros(len(observations), np.float32))
        return self._scalar(self.engine.eval_surr_kl()[1])

    def flat_vpg(self, observations, actions, advantages):
        """flattened gradient of the CPI surrogate as a float32 ndarray -- batch_reinforce.py:54-58."""
        self._bind(observations, actions, advantages)
        return self.engine.to_host(self.engine.surr_vpg()[0])

    # ------------------------------------------------------------------ main loop step
    def train_step(self, N, env=None, sample_mode='trajectories', horizon=1e6, gamma=0.995, gae_lambda=0.97,
                   num_cpu='max', env_kwargs=None):
        """batch_reinforce.py:61-114"""
        if env is None:
            env = self.env.env_id if hasattr(self.env, "env_id") else self.env
        if sample_mode not in ('trajectories', 'samples'):
            raise ValueError("sample_mode must be either 'trajectories' or 'samples'")
        from ..utils import ingest as _ingest
        _ingest.tune_malloc()                   # this is a training process: host allocator policy for rollout batches (once; utils/ingest.py)
        t0 = timer.time()
        # One process per GPU (torch.distributed initialised): N is the size of the WHOLE batch and every rank samples its
        # contiguous share of it.  In sample_mode 'trajectories' the share is a range of EPISODES seeded as a single process
        # would have seeded them (base_seed + episode index, samplers/core.py:44-57): with a sampler that returns exactly the
        # episodes asked for (num_cpu dividing the share -- core.py:124 rounds each worker's count UP otherwise) the ranks' path
        # lists, in rank order, ARE the batch of the one-process run.  In 'samples' mode the share is a number of TIMESTEPS and
        # the seeds are offsets into nothing a single process would have drawn: the ranks still hold disjoint, reproducible
        # batches of the right total size, but not the one-process batch.  Everything after sampling sums over
        # the ranks (returns statistics, advantage whitening, the update, ONE baseline fit).
        n_mine, seed_mine = N, self.seed
        d = _dist()
        if d is not None:
            W, r = d.get_world_size(), d.get_rank()
            lo, hi = r * N // W, (r + 1) * N // W
            n_mine = hi - lo
            seed_mine = None if self.seed is None else self.seed + lo
        common = dict(env=env, policy=self.policy, horizon=horizon, base_seed=seed_mine, num_cpu=num_cpu, env_kwargs=env_kwargs)
        if n_mine <= 0:
            paths = []
        elif sample_mode == 'trajectories':
            # ingestion under sampling: this package's sampler hands every finished chunk of episodes to the stager
            # while the later ones are still being simulated -- rewards, observations and actions are resident when sampling ends
            # (utils/ingest.StreamedBatch; MJX_STREAM_INGEST=0: stage after sampling).  An env ID goes to the reference's own
            # sampler, whose workers return their whole share at once (samplers/core.py:196-205): today's path.
            sink = None if isinstance(env, str) else _ingest.StreamedBatch.for_current_device()
            paths = trajectory_sampler.sample_paths(num_traj=n_mine, **common, **({} if sink is None else {"sink": sink}))
            self.last_ingest = None if sink is None else dict(streamed=bool(sink.finish(paths)), chunks=sink.chunks, why_not=sink.why)
            if d is not None and len(paths) != n_mine:
                # (a sampler that rounds each worker's share up -- core.py:124 with num_cpu not dividing the share -- hands back more
                #  episodes than asked for: the ranks' lists are then no longer the one-process batch)
                import warnings
                warnings.warn("mjrl_amd: rank %d asked its sampler for %d trajectories and got %d (num_cpu = %r does not divide the "
                              "share?): the ranks' batches no longer add up to the one-process batch" % (d.get_rank(), n_mine, len(paths), num_cpu))
        else:
            paths = trajectory_sampler.sample_data_batch(num_samples=n_mine, **common)
        if self.save_logs:
            self.logger.log_kv('time_sampling', timer.time() - t0)
        if self.seed is not None:
            self.seed = self.seed + N

        from ..utils.ingest import drop_shared_batch, trusted_iteration
        with trusted_iteration():               # nothing but this package touches `paths` from here to the baseline fit
            process_samples.compute_returns(paths, gamma)
            # the MLP baseline's epoch permutations -- all that is left of its fit on the critical path -- start now, on a helper
            # thread under the advantage computation and the update (speculative: the fit takes them over only if nobody touched NumPy's global
            # stream in between; MLPBaseline.predraw).  Only for this package's own train_from_paths without row subsampling
            # (which draws from the same stream), one process.
            pre = None
            if (os.environ.get("MJX_ASYNC_FIT", "1") != "0" and hasattr(self.baseline, "predraw") and d is None
                    and (type(self).train_from_paths.__module__ or "").startswith("mjrl_amd.")
                    and not (getattr(self, "hvp_subsample", None) is not None and self.hvp_subsample < 0.99)
                    and type(self).__name__ not in ("PPO",)):
                pre = self.baseline.predraw(sum(len(p["rewards"]) for p in paths))
            process_samples.compute_advantages(paths, self.baseline, gamma, gae_lambda)
            eval_statistics = self.train_from_paths(paths)
            eval_statistics.append(N)
            # The fitted baseline is not read before the NEXT iteration's compute_advantages, and sampling comes first
            # (batch_reinforce.py:78-112): a baseline that can fit in the background (MLPBaseline.fit_async: 0.6 s of a one-
            # workgroup Adam chain per 1M timesteps) is started here and left running under the next rollouts; its errors and
            # duration enter the log as pending entries that turn into numbers when the fit is over (utils/logger.py).
            fit_async = getattr(self.baseline, "fit_async", None) if os.environ.get("MJX_ASYNC_FIT", "1") != "0" else None
            if self.save_logs:
                self.logger.log_kv('num_samples', self.engine.global_count(int(np.sum([p["rewards"].shape[0] for p in paths]))))
                t0 = timer.time()
                if fit_async is not None:
                    self._log_fit_when_done(fit_async(paths, return_errors=True, **({"predrawn": pre} if pre is not None else {})), t0)
                else:
                    error_before, error_after = self.baseline.fit(paths, return_errors=True)
                    self.logger.log_kv('time_VF', timer.time() - t0)
                    self.logger.log_kv('VF_error_before', error_before)
                    self.logger.log_kv('VF_error_after', error_after)
            elif fit_async is not None:
                fit_async(paths, **({"predrawn": pre} if pre is not None else {}))
            else:
                self.baseline.fit(paths)
            if self.save_logs and hasattr(self.baseline, "predraw"):
                # how often the speculative permutation draws were (not) the fit's own draws, cumulative (MLPBaseline._take_predrawn)
                taken, discarded = getattr(self.baseline, "predraw_stats", (0, 0)) if "predraw_stats" in vars(self.baseline) else (0, 0)
                self.logger.log_kv('VF_predraw_taken', taken)
                self.logger.log_kv('VF_predraw_discarded', discarded)
        drop_shared_batch()                     # the iteration's one upload served predict, update and fit; nothing may outlive it
        return eval_statistics

    def _log_fit_when_done(self, pend, t0):
        """time_VF / VF_error_before / VF_error_after of a fit that is still running: pending log entries, delivered by the
        fit's settlement (whoever touches the baseline next, or save_log) and then REPLACED in the log by plain floats"""
        from ..utils.logger import PendingValue
        enqueue_s = timer.time() - t0
        slots = {}
        for key in ('time_VF', 'VF_error_before', 'VF_error_after'):
            slots[key] = PendingValue(pend)
            self.logger.log_kv(key, slots[key])
        log = self.logger.log
        where = {key: len(log[key]) - 1 for key in slots}

        def deliver(errors, device_ms):
            vals = dict(time_VF=enqueue_s + 1e-3 * device_ms, VF_error_before=errors[0], VF_error_after=errors[1])
            for key, ph in slots.items():
                ph.deliver(vals[key])
                series = self.logger.log.get(key)
                if series is not None and where[key] < len(series) and series[where[key]] is ph:
                    series[where[key]] = ph.value
        pend.hooks.append(deliver)
        if pend.done:                                    # (nothing was in flight: a baseline whose fit_async is synchronous)
            deliver(pend.value, pend.device_ms or 0.0)

    # ------------------------------------------------------------------ vanilla PG update
    def train_from_paths(self, paths):
        """batch_reinforce.py:117-175 (optional halving line search on the KL)."""
        base_stats = self._process_and_bind(paths)
        if self.save_logs:
            self.log_rollout_statistics(paths)
        eng = self.engine
        t0 = timer.time()
        g, surr_before = eng.surr_vpg()
        eng.x.copy_(g)
        t_gLL = timer.time() - t0
        alpha = self.alpha
        eng.apply_step(alpha, self.policy.min_log_std)
        if self.desired_kl is not None:
            for _ in range(100):
                if eng.eval_surr_kl()[1] <= self.desired_kl:
                    break
                print("backtracking")
                alpha = alpha / 2.0
                eng.apply_step(alpha, self.policy.min_log_std)
        surr_after, kl_dist = eng.eval_surr_kl()
        self.policy.set_param_values(eng.to_host(eng.theta_new), set_new=True, set_old=True)
        if self.save_logs:
            self.logger.log_kv('alpha', self.alpha)
            self.logger.log_kv('time_vpg', t_gLL)
            self.logger.log_kv('kl_dist', kl_dist)
            self.logger.log_kv('surr_improvement', surr_after - surr_before)
            self.logger.log_kv('running_score', self.running_score)
            self._log_success(paths)
        return base_stats

    # ------------------------------------------------------------------ path bookkeeping
    def process_paths(self, paths):
        """batch_reinforce.py:178-197: concatenate, whiten advantages (population std + 1e-6),
        return statistics, running score.  With torch.distributed initialised, `paths` is this
        rank's trajectory shard and mean / std / return statistics are taken over all ranks."""
        observations = np.concatenate([path["observations"] for path in paths])
        actions = np.concatenate([path["actions"] for path in paths])
        advantages, base_stats, running_score = self._advantages_and_statistics(paths)
        return observations, actions, advantages, base_stats, running_score

    def _path_statistics(self, paths):
        """[mean, std, min, max] of the path returns over all ranks and the running score (batch_reinforce.py:187-196)"""
        path_returns = None
        eng = self._engine_obj
        if eng is not None and eng.device.type == "cuda" and len(paths) > 64:
            # the rewards were staged back to back in a page-locked block by compute_returns: one vectorised segmented sum
            # instead of a NumPy call per path (4 ms per 1 000 paths)
            from ..utils import ingest
            blk = ingest.host_block(eng.backend, paths, "rewards")
            if blk is not None:
                lens = np.fromiter((len(p["rewards"]) for p in paths), dtype=np.int64, count=len(paths))
                if lens.min() > 0:
                    starts = np.zeros(len(paths), np.int64)
                    np.cumsum(lens[:-1], out=starts[1:])
                    path_returns = np.add.reduceat(blk.reshape(-1), starts)
        if path_returns is None and eng is not None and getattr(eng, "lib", None) is not None and len(paths) > 16:
            # host arrays as the sampler left them: addresses by one C walk over the path list (csrc/pathwalk.c), the sums on
            # libmjx's host threads -- each path added left to right like the reference's sum(p["rewards"]) (its very bits);
            # a NumPy call per path cost 3 ms per 1 000 paths, most of this thread's share of train_from_paths
            from ..utils import ingest
            got = ingest.collect_arrays(paths, "rewards")
            if got is not None and got[2] == 1 and got[3] == 8:
                import ctypes
                from .._lib import check
                path_returns = np.empty(len(paths), np.float64)
                check(eng.lib.mjx_host_segment_sums(ctypes.c_void_p(got[0].ctypes.data), ctypes.c_void_p(got[1].ctypes.data),
                                                    len(paths), ctypes.c_void_p(path_returns.ctypes.data), 16))
        if path_returns is None:
            add = np.add.reduce                      # (np.sum's dispatch wrappers cost more than a 1 000-element sum: 3.0 -> 1.3 us per path)
            path_returns = np.fromiter((float(add(p["rewards"])) for p in paths), dtype=np.float64, count=len(paths))
        d = _dist()
        if d is not None:
            gathered = [None] * d.get_world_size()
            d.all_gather_object(gathered, path_returns)
            path_returns = np.concatenate(gathered)
        mean_return = np.mean(path_returns)
        base_stats = [mean_return, np.std(path_returns), np.amin(path_returns), np.amax(path_returns)]
        running_score = mean_return if self.running_score is None else 0.9 * self.running_score + 0.1 * mean_return
        return base_stats, running_score

    def _advantages_and_statistics(self, paths):
        advantages = np.concatenate([path["advantages"] for path in paths])
        mean, std = self._global_mean_std(advantages)
        advantages = (advantages - mean) / (std + 1e-6)
        base_stats, running_score = self._path_statistics(paths)
        return advantages, base_stats, running_score

    def _staging_pool(self):
        if getattr(self, "_stage_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._stage_pool = ThreadPoolExecutor(max_workers=1)
        return self._stage_pool

    def _stage_on_callers_stream(self):
        """eng.stage_paths wrapped for a helper thread: torch's current device / stream are thread-local, so the worker
        adopts the CALLER's (the stager orders the caller's stream after its side-stream transfers, and page-locked
        allocations must not initialise a context on GPU 0 from a rank that owns another GPU)."""
        eng = self.engine
        torch, dev = eng.torch, eng.device
        from ..utils.ingest import carried_trust
        trust = carried_trust()                 # (thread-local too: inside train_step the worker re-uses the staged batch by identity)
        if dev.type != "cuda":
            def run_cpu(paths, keys):
                with trust():
                    return eng.stage_paths(paths, keys)
            return run_cpu
        cur = torch.cuda.current_stream(dev)

        def run(paths, keys):
            with torch.cuda.device(dev), torch.cuda.stream(cur), trust():
                return eng.stage_paths(paths, keys)
        return run

    def _process_and_bind(self, paths, defer_stats=False):
        """process_paths + upload + binding for train_from_paths: the (whitened, fp64) advantages are assembled on the
        host like the reference does, observations / actions go path by path through the engine's page-locked
        stager (utils/ingest.py) -- no concatenated host copy, transfers overlapped with staging.
        -> base_stats; sets self.running_score.  defer_stats (r06): -> a callable that returns base_stats (and sets the running
        score) when asked -- the per-path return statistics are host work the update does not depend on; NPG runs them under the
        update's device time."""
        # the gather / upload of observations and actions runs on a helper thread (native memcpy threads + asynchronous
        # copies, no GIL) while this thread assembles the advantage vector and the path statistics
        eng = self.engine
        from ..utils import ingest
        if not paths:
            bs = self._bind_empty_shard()
            return (lambda: bs) if defer_stats else bs
        # (asked before the staging jobs start: a helper holds its key's registry lock while it stages)
        adv64 = ingest.lookup(eng.backend, paths, "advantages") if eng.device.type == "cuda" else None
        host_adv = False
        if _dist() is not None:
            # which route the advantages take decides which collectives follow (engine.whitened_advantages: libmjx's transport;
            # _advantages_and_statistics: torch.distributed): the ranks must agree on it.  The device block is used only when
            # EVERY rank holds one; the host-array fast path below depends on per-rank path counts and is left to single
            # processes.
            from ..utils import ranks
            if not ranks.all_true(adv64 is not None):
                adv64 = None
        elif adv64 is None and eng.device.type == "cuda" and len(paths) > 64 and getattr(eng, "_stager", None) is None:
            got = ingest.collect_arrays(paths, "advantages")             # one C walk: uniform 1-D float64 arrays?
            host_adv = (got is not None and got[2] == 1 and got[3] == 8 and isinstance(paths[0]["advantages"], np.ndarray)
                        and paths[0]["advantages"].ndim == 1) if ingest._pathwalk is not None else all(
                isinstance(p["advantages"], np.ndarray) and p["advantages"].ndim == 1 and p["advantages"].dtype == np.float64 for p in paths)
        # observations and actions start their way to the device NOW: one native staging job per block (mjx_stage_async: libmjx's
        # threads gather / convert the trajectories group by group and queue each group's copy behind it; no Python thread, no
        # interpreter lock) -- the call returns at once and this thread goes on with the advantages and the path statistics.  The
        # update cannot begin until the last block has landed; everything else fits under that.
        native = getattr(eng, "_stager", None) is None and eng.device.type == "cuda"
        futs, staged = [], {}
        if native:
            # the policy's four small (synchronous, pageable) uploads go FIRST, while no copy engine is busy: issued under the
            # staging jobs -- as r04-r06 did -- they wait for a DMA engine whenever the runtime hands them the one that is moving
            # the 90 MB observation / action blocks: 6.7 ms of `set_policy` in every other call of a process's first ten
            # (`tools/e2e_timeline.py TL_FIRST=1`, profiles/r06b_bench/e2e_first_calls.log), 0.1 ms here
            self._push_policy()
            staged = eng.stage_paths(paths, ("observations", "actions"), defer=True)
        else:                                    # (CPU stand-ins, caller-supplied stagers: a helper thread)
            futs = [self._staging_pool().submit(self._stage_on_callers_stream(), paths, ("observations", "actions"))]
        try:
            if host_adv:
                # host advantages (train_from_paths called on its own): the 8 bytes per timestep go up through the stager as they are
                # and are whitened on the device like the resident ones -- np.concatenate + mean + std + the division cost 2-3 ms of
                # this thread per 1M timesteps
                adv64 = ingest.stage_shared(eng.backend, paths, ("advantages",))["advantages"]["raw"].view(-1)
            stats_later = None
            if adv64 is not None:
                # the advantages never left the device (utils/process_samples.compute_advantages): whitening statistics
                # and the fp32 cast happen there; only the per-path return statistics are host work
                advantages = eng.whitened_advantages(adv64)
                if defer_stats:
                    def stats_later():
                        bs, self.running_score = self._path_statistics(paths)
                        return bs
                else:
                    base_stats, self.running_score = self._path_statistics(paths)
            else:
                advantages, base_stats, self.running_score = self._advantages_and_statistics(paths)
        finally:
            for f in futs:
                staged.update(f.result())
            if native:
                ingest.settle(eng.backend)       # the staging jobs have queued their copies; this stream waits for them
        if not native:
            self._push_policy()
        eng.set_batch(staged["observations"], staged["actions"], advantages)
        if defer_stats:
            return stats_later if stats_later is not None else (lambda: base_stats)
        return base_stats

    def _bind_empty_shard(self):
        """a rank that holds no trajectories this iteration (more ranks than paths): it binds zero rows and still takes part in
        every rank sum -- statistics, gradient, Fisher products -- so the other ranks' update is the one-process update"""
        eng = self.engine
        from ..utils import ranks
        if _dist() is not None:
            ranks.all_true(False)                        # the others' "does every rank hold a device block" round: no
        mean, std = self._global_mean_std(np.zeros(0))
        base_stats, self.running_score = self._path_statistics([])
        self._push_policy()
        z = np.zeros
        eng.set_batch(z((0, self.policy.n), np.float32), z((0, self.policy.m), np.float32), z(0, np.float32))
        return base_stats

    def _global_mean_std(self, x):
        """population mean / std of a sample vector that is sharded over the ranks"""
        if _dist() is None:
            return np.mean(x), np.std(x)
        from ..utils import ranks
        return ranks.mean_std(x)

    def _global_column_mean_std(self, X):
        """column-wise population mean / std of a (rows, n) block that is sharded over the ranks (the observation
        statistics of npg_cg.py:104-105 must be the same on every rank, or the ranks' transforms drift apart)"""
        d = _dist()
        if d is None:
            return np.mean(X, axis=0), np.std(X, axis=0)
        torch = self.engine.torch
        s = torch.from_numpy(np.concatenate([X.sum(axis=0, dtype=np.float64), [float(X.shape[0])]])).to(self.engine.device)
        d.all_reduce(s)
        cnt = float(s[-1].item())
        mean = s[:-1].cpu().numpy() / cnt
        q = torch.from_numpy(((X - mean) ** 2).sum(axis=0, dtype=np.float64)).to(self.engine.device)
        d.all_reduce(q)
        return mean, np.sqrt(q.cpu().numpy() / cnt)

    def log_rollout_statistics(self, paths):
        """batch_reinforce.py:200-214 (over the paths of all ranks, like every other logged statistic)"""
        path_returns = [float(np.sum(p["rewards"])) for p in paths]
        d = _dist()
        if d is not None:
            gathered = [None] * d.get_world_size()
            d.all_gather_object(gathered, path_returns)
            path_returns = [r for g in gathered for r in g]
        self.logger.log_kv('stoc_pol_mean', np.mean(path_returns))
        self.logger.log_kv('stoc_pol_std', np.std(path_returns))
        self.logger.log_kv('stoc_pol_max', np.amax(path_returns))
        self.logger.log_kv('stoc_pol_min', np.amin(path_returns))
        try:
            self.logger.log_kv('rollout_success', self.env.env.env.evaluate_success(paths))
        except Exception:
            pass

    def _log_success(self, paths):
        try:
            self.env.env.env.evaluate_success(paths, self.logger)
        except Exception:
            try:
                self.logger.log_kv('success_rate', self.env.env.env.evaluate_success(paths))
            except Exception:
                pass
