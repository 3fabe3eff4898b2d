"""Rollout collection stays on host CPUs (north star; reference mjrl/samplers/core.py).

When mjrl is installed its own sampler is used unchanged (our Policy is picklable and its
``get_action`` is NumPy-only, so it travels into mjrl's worker pool).  Otherwise a minimal
serial sampler with the same path format is provided so that ``train_step`` works against
any env object exposing ``reset() / step(a) / horizon`` (used by the tests' NumPy envs).
"""
import numpy as np

try:                                                    # pragma: no cover - depends on the host env
    from mjrl.samplers.core import sample_paths, sample_data_batch  # noqa: F401
    HAVE_MJRL = True
except Exception:
    HAVE_MJRL = False

    def _make_env(env, env_kwargs):
        if callable(env) and not hasattr(env, "step"):
            return env(**(env_kwargs or {}))
        if isinstance(env, str):
            raise RuntimeError("string env ids need mjrl + gym on the host; pass an env object or factory")
        return env

    def sample_paths(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1,
                     max_process_time=300, max_timeouts=4, suppress_print=False, env_kwargs=None):
        """Path dict format of mjrl/samplers/core.py:85-93; seeding of :44-57."""
        env = _make_env(env, env_kwargs)
        if base_seed is not None:
            env.set_seed(base_seed) if hasattr(env, "set_seed") else None
            np.random.seed(base_seed)
        T = int(min(horizon, getattr(env, "horizon", horizon)))
        paths = []
        for ep in range(num_traj):
            if base_seed is not None:
                if hasattr(env, "set_seed"):
                    env.set_seed(base_seed + ep)
                np.random.seed(base_seed + ep)
            obs, acts, rews = [], [], []
            o, done, t = env.reset(), False, 0
            while t < T and not done:
                a, info = policy.get_action(o)
                if eval_mode:
                    a = info['evaluation']
                nxt, r, done, _ = env.step(a)
                obs.append(o); acts.append(a); rews.append(r)
                o, t = nxt, t + 1
            paths.append(dict(observations=np.array(obs), actions=np.array(acts), rewards=np.array(rews),
                              agent_infos={}, env_infos={}, terminated=bool(done)))
        return paths

    def sample_data_batch(num_samples, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1,
                          paths_per_call=1, env_kwargs=None):
        paths, got, seed = [], 0, base_seed
        while got < num_samples:
            new = sample_paths(paths_per_call, env, policy, eval_mode, horizon, seed, 1, env_kwargs=env_kwargs)
            paths += new
            got += sum(len(p["rewards"]) for p in new)
            seed = None if seed is None else seed + paths_per_call
        return paths
