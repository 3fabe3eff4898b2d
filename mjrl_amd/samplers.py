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

What spawn asks of the caller (r06): a spawned worker imports the training script again as ``__mp_main__`` and finds
env factories / policy classes BY NAME.  (i) A script without an ``if __name__ == "__main__":`` guard would re-run itself in
every worker: such a main module is detected and the request is served in this process (one warning; guard the script or set
``MJX_SAMPLER_START=fork``).  (ii) An env factory or policy class the workers cannot import (defined in ``__main__`` of a script
run through ``runpy``, a lambda, a local class) fails INSIDE the worker's job -- the payload travels as bytes and is unpickled
there -- and comes back as an error at once instead of a lost task and a 300 s timeout: same fall-back, same warning.
(iii) ``num_cpu='max'`` means ``min(cpu_count, MJX_SAMPLER_MAX_WORKERS = 32)`` interpreters, not 256 on a 256-thread host.

Streaming: with a ``sink`` (utils/ingest.StreamedBatch -- train_step passes one) the request is cut into
more jobs than workers, the results are taken in EPISODE ORDER as they arrive and every chunk's rewards / observations /
actions are gathered into the page-locked staging blocks and sent to the GPU while the later episodes are still being
simulated: when sampling ends the batch is resident.  Same episodes, same seeds (base_seed + episode index), same path list.
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
    if isinstance(env, type) or (callable(env) and not hasattr(env, "step")):
        # a factory -- or an env CLASS (core.py:36-39 instantiates any callable; a class has `step` too, as a plain function)
        return env(**(env_kwargs or {}))
    if hasattr(env, "step") and hasattr(env, "reset"):
        return env
    print("Unsupported environment format")
    raise AttributeError


def native_do_rollout(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, env_kwargs=None, _env_obj=None):
    """mjrl/samplers/core.py:13-97 -- one process, `num_traj` episodes, episode `ep` seeded with base_seed + ep
    (_env_obj: an env built from `env` earlier -- the streaming jobs of one worker share it like the episodes of one job do)"""
    env = _env_obj if _env_obj is not None else _make_env(env, env_kwargs)
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
_SERIAL_ONLY = {}            # why this process serves num_cpu > 1 requests itself (set once, with the warning): reason string


def _start_method():
    m = os.environ.get("MJX_SAMPLER_START", "spawn")
    if m not in ("spawn", "forkserver", "fork"):
        raise ValueError("MJX_SAMPLER_START must be spawn, forkserver or fork, not %r" % m)
    return m


def _resolve_num_cpu(num_cpu):
    """core.py:113-115; 'max' is capped (MJX_SAMPLER_MAX_WORKERS, default 32): every worker is a spawned interpreter"""
    num_cpu = 1 if num_cpu is None else num_cpu
    if num_cpu == 'max':
        num_cpu = max(1, min(mp.cpu_count(), int(os.environ.get("MJX_SAMPLER_MAX_WORKERS", "32"))))
    assert type(num_cpu) == int                                                   # noqa: E721 (core.py:115)
    return num_cpu


def _main_would_rerun():
    """would a spawned worker, importing this process's main module again, re-run a training script?  (a main module that is a
    file without an `if __name__ == "__main__":` guard; `python -c`, interactive sessions and guarded scripts: no)"""
    import re
    import sys
    main = sys.modules.get("__main__")
    spec = getattr(main, "__spec__", None)
    path = getattr(spec, "origin", None) if getattr(spec, "name", None) else getattr(main, "__file__", None)
    if not path or not os.path.isfile(path) or not str(path).endswith(".py"):
        return False
    try:
        with open(path, errors="replace") as f:
            src = f.read()
    except OSError:                                      # pragma: no cover
        return False
    return re.search(r"^[ \t]*if[ \t]+__name__[ \t]*==[ \t]*['\"]__main__['\"]", src, re.M) is None


def _serve_here(reason, sticky=True):
    """say (once per process) that a num_cpu > 1 request is served in this process; sticky: ... and every later one too (a property
    of the PROCESS -- its main module; a payload that cannot travel concerns that request only)"""
    if sticky:
        _SERIAL_ONLY.setdefault("why", reason)
    if "warned" not in _SERIAL_ONLY:
        _SERIAL_ONLY["warned"] = True
        import warnings
        warnings.warn("mjrl_amd.samplers: num_cpu > 1 is served in the training process itself -- %s.  Guard the script with "
                      "`if __name__ == \"__main__\":` and define env factories / policy classes in an importable module, or set "
                      "MJX_SAMPLER_START=fork (forked workers inherit everything; libmjx refuses device work in them and they ask for none)." % reason)


class _PayloadError(Exception):
    """a worker could not rebuild the job's env / policy from the bytes it was sent"""


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

_WORKER = {}                 # worker-side: the last env / policy payloads and what they unpickled to


def _rollout_job(env_blob, policy_blob, num_traj, base_seed, eval_mode, horizon):
    """runs in a worker: (env, env_kwargs) and the policy arrive as BYTES and are rebuilt here, so that a payload the worker cannot
    import fails as an ordinary exception of this job (-> _PayloadError) instead of killing the worker while it reads its task
    queue.  The env built from `env_blob` is kept between jobs (like the episodes of one reference job share theirs: every episode
    is re-seeded, core.py:57-60); the policy is rebuilt whenever its bytes change (once per iteration).  -> (paths, T)"""
    import pickle
    try:
        if _WORKER.get("env_blob") != env_blob:
            env, env_kwargs = pickle.loads(env_blob)
            _WORKER.update(env_blob=env_blob, env=_make_env(env, env_kwargs))
        if _WORKER.get("policy_blob") != policy_blob:
            _WORKER.update(policy_blob=policy_blob, policy=pickle.loads(policy_blob))
    except Exception as e:
        _WORKER.clear()
        raise _PayloadError("%s: %s" % (type(e).__name__, e))
    env = _WORKER["env"]
    paths = native_do_rollout(num_traj, None, _WORKER["policy"], eval_mode, horizon, base_seed, None, _env_obj=env)
    return paths, min(horizon, getattr(env, "horizon", horizon))


def _try_multiprocess(jobs, num_cpu, max_process_time, max_timeouts, sink=None):
    """core.py:189-210 on the persistent pool: all jobs or nothing; a timeout tears the workers down and retries.  The results are
    taken in job (= episode) order; with a sink every chunk is handed on the moment it is there, while later jobs still run."""
    for attempt in range(int(max_timeouts)):
        pool = _pool(num_cpu)
        runs = [pool.apply_async(_rollout_job, args=j) for j in jobs]
        out = []
        try:
            for r in runs:
                paths, T = r.get(timeout=max_process_time)
                out.append(paths)
                if sink is not None:
                    sink.add(paths, T)
            return out
        except mp.TimeoutError as e:
            print(str(e))
            print("Timeout Error raised... Trying again")
            _drop_pool(num_cpu)
            if sink is not None:
                sink.abort("a sampler timeout: the request was retried")      # (what was streamed belongs to a discarded attempt)
        except Exception:
            _drop_pool(num_cpu)
            raise
    return None


def native_sample_paths(num_traj, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1,
                        max_process_time=300, max_timeouts=4, suppress_print=False, env_kwargs=None, sink=None):
    """mjrl/samplers/core.py:99-148: num_cpu == 1 in this process; otherwise ceil(num_traj / num_cpu) episodes per worker,
    worker i seeded base_seed + i * paths_per_cpu -- the episodes (and, when num_cpu divides num_traj, their order) of the
    one-process call.  sink (utils/ingest.StreamedBatch): see the module docstring -- every worker's share is cut into
    MJX_SAMPLER_PIECES (4) jobs, each seeded base_seed + its first episode's index, so the episodes are the same ones."""
    import pickle
    num_cpu = _resolve_num_cpu(num_cpu)
    common = dict(env=env, policy=policy, eval_mode=eval_mode, horizon=horizon, env_kwargs=env_kwargs)
    pieces = max(1, int(os.environ.get("MJX_SAMPLER_PIECES", "4"))) if sink is not None else 1
    if num_cpu > 1 and "why" not in _SERIAL_ONLY and _start_method() != "fork" and _main_would_rerun():
        _serve_here("the main module of this process is a script without a __main__ guard, which every spawned worker would run again")
    blobs = None
    if num_cpu > 1 and "why" not in _SERIAL_ONLY:
        try:
            blobs = (pickle.dumps((env, env_kwargs)), pickle.dumps(policy))
        except Exception as e:                            # a lambda, a local class, an env holding an open handle ...
            _serve_here("the env / policy cannot be pickled for the workers (%s: %s)" % (type(e).__name__, e), sticky=False)
    if blobs is None:
        # this process: the reference's one-process call (episodes base_seed + ep); with a sink in pieces, each handed on when done
        total = num_traj if num_cpu == 1 else num_cpu * int(np.ceil(num_traj / num_cpu))       # (core.py:124 rounds every worker's share up)
        if sink is None:
            return native_do_rollout(num_traj=total, base_seed=base_seed, **common)
        env_obj = _make_env(env, env_kwargs)
        T = min(horizon, getattr(env_obj, "horizon", horizon))
        sink.begin(total)
        step = max(1, int(np.ceil(total / (4 * pieces))))
        paths = []
        for lo in range(0, total, step):
            chunk = native_do_rollout(min(step, total - lo), None, policy, eval_mode, horizon,
                                      None if base_seed is None else base_seed + lo, None, _env_obj=env_obj)
            paths += chunk
            sink.add(chunk, T)
        return paths
    paths_per_cpu = int(np.ceil(num_traj / num_cpu))
    piece = max(1, int(np.ceil(paths_per_cpu / pieces)))
    jobs = []
    for i in range(num_cpu):
        for lo in range(0, paths_per_cpu, piece):
            ep = i * paths_per_cpu + lo
            jobs.append(blobs + (min(piece, paths_per_cpu - lo), None if base_seed is None else base_seed + ep, eval_mode, horizon))
    if suppress_print is False:
        start_time = timer.time()
        print("####### Gathering Samples #######")
    if sink is not None:
        sink.begin(num_cpu * paths_per_cpu)
    try:
        results = _try_multiprocess(jobs, num_cpu, max_process_time, max_timeouts, sink)
    except _PayloadError as e:
        _serve_here("the workers cannot rebuild the env / policy they were sent (%s)" % e, sticky=False)
        if sink is not None:
            sink.abort("the worker pool was given up")
        total = num_cpu * paths_per_cpu                                     # (the episodes the pool would have returned, seeded alike)
        return native_do_rollout(num_traj=total, base_seed=base_seed, **common)
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
    num_cpu = _resolve_num_cpu(num_cpu)
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
                 max_timeouts=4, suppress_print=False, env_kwargs=None, sink=None):
    """An env ID (what train_step passes when agent.env is mjrl's GymEnv, batch_reinforce.py:71) can only be resolved by mjrl's own
    GymEnv / gym.make: such calls go to the UNMODIFIED mjrl.samplers.core.sample_paths, fork pool and all (a `sink` is not
    used: the reference's workers hand back their whole share at once, core.py:196-205).  Env objects and factories are served here."""
    if _is_env_id(env):
        return _mjrl_core().sample_paths(num_traj, env, policy, eval_mode, horizon, base_seed, num_cpu, max_process_time,
                                         max_timeouts, suppress_print, env_kwargs)
    return native_sample_paths(num_traj, env, policy, eval_mode, horizon, base_seed, num_cpu, max_process_time, max_timeouts,
                               suppress_print, env_kwargs, sink)


def sample_data_batch(num_samples, env, policy, eval_mode=False, horizon=1e6, base_seed=None, num_cpu=1, paths_per_call=1,
                      env_kwargs=None):
    if _is_env_id(env):
        return _mjrl_core().sample_data_batch(num_samples, env, policy, eval_mode, horizon, base_seed, num_cpu, paths_per_call,
                                              env_kwargs)
    return native_sample_data_batch(num_samples, env, policy, eval_mode, horizon, base_seed, num_cpu, paths_per_call, env_kwargs)
