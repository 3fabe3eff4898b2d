"""Rollout collection stays on host CPUs (north star; reference mjrl/samplers/core.py).

An env given as a string ID belongs to mjrl's own GymEnv / gym.make: those calls go to the UNMODIFIED
``mjrl.samplers.core`` -- our Policy is picklable and its ``get_action`` is NumPy-only, so it travels into mjrl's forked
worker pool (core.py:189-210); libmjx refuses device work in such a child (``mjx_process_state``, include/mjx.h) and the
workers never ask for any.  For env objects and factories this module provides the same three functions -- ``do_rollout`` (core.py:13-97), ``sample_paths`` (:99-148), ``sample_data_batch`` (:151-186) -- with
the reference's path format, per-episode seeding and per-worker split, so that ``train_step`` works against any env
object or factory exposing ``reset() / step(a) / horizon`` (the tests' NumPy envs).

``num_cpu > 1`` is honoured by a pool of SPAWNED workers, kept alive between calls (a spawned interpreter inherits nothing
from the training process: no HIP context, no page-locked blocks, no half-held locks of libmjx's gather threads; what it
costs -- an interpreter start per worker -- is paid once per job instead of once per iteration).  ``MJX_SAMPLER_START``
= ``spawn`` (default) | ``forkserver`` | ``fork`` picks the start method.  ``max_process_time`` / ``max_timeouts`` act like
the reference's: a worker set that does not answer in time is torn down and the whole request is retried.
"""
import atexit
import multiprocessing as mp
import os
import time as timer

import numpy as np


def _stack_dict_list(dicts):
    """utils/tensor_utils.py:75-91: a list of {key: value or nested dict} -> {key: stacked array or nested dict}"""
    if not dicts:
        return {}
    out = {}
    for k in dicts[0].keys():
        example = dicts[0][k]
        if isinstance(example, dict):
            out[k] = _stack_dict_list([x[k] for x in dicts])
        else:
            out[k] = np.array([x[k] for x in dicts])
    return out


def _make_env(env, env_kwargs):
    if isinstance(env, str):
        raise RuntimeError("string env ids need mjrl + gym on the host (mjrl/utils/gym_env.py:23-24); pass an env object or a factory")
    if callable(env) and not hasattr(env, "step"):
        return env(**(env_kwargs or {}))
    if hasattr(env, "step") and hasattr(env, "reset"):
        return env
    print("Unsupported environment format")
    raise AttributeError


def native_do_rollout(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, env_kwargs=None):
    """mjrl/samplers/core.py:13-97 -- one process, `num_traj` episodes, episode `ep` seeded with base_seed + ep"""
    env = _make_env(env, env_kwargs)
    seed_env = getattr(env, "set_seed", None)
    if base_seed is not None:
        if seed_env is not None:
            seed_env(base_seed)
        np.random.seed(base_seed)
    else:
        np.random.seed()
    T = min(horizon, getattr(env, "horizon", horizon))
    infos_of = getattr(env, "get_env_infos", None)
    paths = []
    for ep in range(num_traj):
        if base_seed is not None:
            seed = base_seed + ep
            if seed_env is not None:
                seed_env(seed)
            np.random.seed(seed)
        obs, acts, rews, agent_infos, env_infos = [], [], [], [], []
        o, done, t = env.reset(), False, 0
        while t < T and done != True:                                      # noqa: E712  (the reference's own test, core.py:68)
            a, agent_info = policy.get_action(o)
            if eval_mode:
                a = agent_info['evaluation']
            env_info_base = infos_of() if infos_of is not None else {}
            nxt, r, done, env_info_step = env.step(a)
            env_info = env_info_step if env_info_base == {} else env_info_base
            obs.append(o); acts.append(a); rews.append(r); agent_infos.append(agent_info); env_infos.append(env_info or {})
            o, t = nxt, t + 1
        paths.append(dict(observations=np.array(obs), actions=np.array(acts), rewards=np.array(rews),
                          agent_infos=_stack_dict_list(agent_infos), env_infos=_stack_dict_list(env_infos), terminated=done))
    del env
    return paths


# ---------------------------------------------------------------------------------------------------------------- worker pool
_POOLS = {}


def _start_method():
    m = os.environ.get("MJX_SAMPLER_START", "spawn")
    if m not in ("spawn", "forkserver", "fork"):
        raise ValueError("MJX_SAMPLER_START must be spawn, forkserver or fork, not %r" % m)
    return m


def _pool(num_cpu):
    key = (int(num_cpu), _start_method())
    p = _POOLS.get(key)
    if p is None:
        p = _POOLS[key] = mp.get_context(key[1]).Pool(processes=key[0])
    return p


def _drop_pool(num_cpu):
    p = _POOLS.pop((int(num_cpu), _start_method()), None)
    if p is not None:
        p.terminate()
        p.join()


def close_pools():
    """tear the worker pools down (also at interpreter exit)"""
    for key in list(_POOLS):
        p = _POOLS.pop(key)
        p.terminate()
        p.join()


atexit.register(close_pools)


def _try_multiprocess(func, input_dict_list, num_cpu, max_process_time, max_timeouts):
    """core.py:189-210 on the persistent pool: all jobs or nothing; a timeout tears the workers down and retries"""
    for _ in range(int(max_timeouts)):
        pool = _pool(num_cpu)
        runs = [pool.apply_async(func, kwds=d) for d in input_dict_list]
        try:
            return [r.get(timeout=max_process_time) for r in runs]
        except mp.TimeoutError as e:
            print(str(e))
            print("Timeout Error raised... Trying again")
            _drop_pool(num_cpu)
        except Exception:
            _drop_pool(num_cpu)
            raise
    return None


def native_sample_paths(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1,
                        max_process_time=300, max_timeouts=4, suppress_print=False, env_kwargs=None):
    """mjrl/samplers/core.py:99-148: num_cpu == 1 in this process; otherwise ceil(num_traj / num_cpu) episodes per worker,
    worker i seeded base_seed + i * paths_per_cpu -- the episodes (and, when num_cpu divides num_traj, their order) of the
    one-process call."""
    num_cpu = 1 if num_cpu is None else num_cpu
    num_cpu = mp.cpu_count() if num_cpu == 'max' else num_cpu
    assert type(num_cpu) == int                                                   # noqa: E721 (core.py:115)
    common = dict(env=env, policy=policy, eval_mode=eval_mode, horizon=horizon, env_kwargs=env_kwargs)
    if num_cpu == 1:
        return native_do_rollout(num_traj=num_traj, base_seed=base_seed, **common)
    paths_per_cpu = int(np.ceil(num_traj / num_cpu))
    jobs = [dict(num_traj=paths_per_cpu, base_seed=None if base_seed is None else base_seed + i * paths_per_cpu, **common)
            for i in range(num_cpu)]
    if suppress_print is False:
        start_time = timer.time()
        print("####### Gathering Samples #######")
    results = _try_multiprocess(native_do_rollout, jobs, num_cpu, max_process_time, max_timeouts)
    if results is None:
        raise RuntimeError("sample_paths: %d worker timeouts of %s s each -- no rollouts (mjrl/samplers/core.py:192-193 returns None here, "
                           "which its caller then fails on)" % (max_timeouts, max_process_time))
    paths = [path for result in results for path in result]
    if suppress_print is False:
        print("======= Samples Gathered  ======= | >>>> Time taken = %f " % (timer.time() - start_time))
    return paths


def native_sample_data_batch(num_samples, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1,
                             paths_per_call=1, env_kwargs=None):
    """mjrl/samplers/core.py:151-186: rounds of paths_per_call * num_cpu episodes until num_samples timesteps are in"""
    num_cpu = 1 if num_cpu is None else num_cpu
    num_cpu = mp.cpu_count() if num_cpu == 'max' else num_cpu
    assert type(num_cpu) == int                                                   # noqa: E721
    start_time = timer.time()
    print("####### Gathering Samples #######")
    sampled_so_far, paths = 0, []
    base_seed = 123 if base_seed is None else base_seed
    while sampled_so_far < num_samples:
        base_seed = base_seed + 12345
        new_paths = native_sample_paths(paths_per_call * num_cpu, env, policy, eval_mode, horizon, base_seed, num_cpu,
                                        suppress_print=True, env_kwargs=env_kwargs)
        paths += new_paths
        sampled_so_far += int(np.sum([len(p['rewards']) for p in new_paths]))
    print("======= Samples Gathered  ======= | >>>> Time taken = %f " % (timer.time() - start_time))
    print("................................. | >>>> # samples = %i # trajectories = %i " % (sampled_so_far, len(paths)))
    return paths


# ---------------------------------------------------------------------------------------------------------------- dispatch
def _mjrl_core():
    try:
        from mjrl.samplers import core
        return core
    except Exception as e:                                   # pragma: no cover - depends on the host env
        raise RuntimeError("env given as a string id (%s): that needs mjrl + gym on this host (mjrl/utils/gym_env.py:23-24); "
                           "pass an env object or a factory instead" % e)


def _is_env_id(env):
    return isinstance(env, str)


def do_rollout(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, env_kwargs=None):
    if _is_env_id(env):
        return _mjrl_core().do_rollout(num_traj, env, policy, eval_mode, horizon, base_seed, env_kwargs)
    return native_do_rollout(num_traj, env, policy, eval_mode, horizon, base_seed, env_kwargs)


def sample_paths(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1, max_process_time=300,
                 max_timeouts=4, suppress_print=False, env_kwargs=None):
    """An env ID (what train_step passes when agent.env is mjrl's GymEnv, batch_reinforce.py:71) can only be resolved by mjrl's own
    GymEnv / gym.make: such calls go to the UNMODIFIED mjrl.samplers.core.sample_paths, fork pool and all.  Env objects and
    factories are served here."""
    if _is_env_id(env):
        return _mjrl_core().sample_paths(num_traj, env, policy, eval_mode, horizon, base_seed, num_cpu, max_process_time,
                                         max_timeouts, suppress_print, env_kwargs)
    return native_sample_paths(num_traj, env, policy, eval_mode, horizon, base_seed, num_cpu, max_process_time, max_timeouts,
                               suppress_print, env_kwargs)


def sample_data_batch(num_samples, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1, paths_per_call=1,
                      env_kwargs=None):
    if _is_env_id(env):
        return _mjrl_core().sample_data_batch(num_samples, env, policy, eval_mode, horizon, base_seed, num_cpu, paths_per_call,
                                              env_kwargs)
    return native_sample_data_batch(num_samples, env, policy, eval_mode, horizon, base_seed, num_cpu, paths_per_call, env_kwargs)
