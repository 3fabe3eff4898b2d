"""A NumPy point-mass behind a stand-in ``gym.make`` (TEST INFRASTRUCTURE; SURVEY 4 plan (3)): what lets the UNMODIFIED
reference driver -- mjrl/utils/train_agent.py:62-155, mjrl/samplers/core.py:99-210, mjrl/utils/gym_env.py:18-60 -- run here and on
the GPU box, where neither gym nor MuJoCo exist.  The env reports, with every step, which process stepped it and what that
process has done with libmjx (``mjx_process_state``): the rollouts themselves carry the proof that sampler workers never
touched the device."""
import os
import sys
import types

import numpy as np

ENV_ID = "mjx_point_mass-v0"


class _Box:
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, np.float64), np.asarray(high, np.float64)
        self.shape = self.low.shape


class _Spec:
    id = ENV_ID
    max_episode_steps = 25


def _libmjx_state():
    """(device-touching libmjx entries made by this process, 1 if forked from a process that had made some, 1 if libmjx is loaded)"""
    mod = sys.modules.get("mjrl_amd._lib")
    lib = getattr(mod, "_lib", None) if mod is not None else None
    if lib is None:
        return 0, 0, 0
    import ctypes
    out = (ctypes.c_int64 * 2)()
    lib.mjx_process_state(out)
    return int(out[0]), int(out[1]), 1


class PointMassGym:
    """obs = [pos(2), vel(2), target(2)], act = force(2) in [-1, 1], 25 steps, done when the mass sits on the target"""
    spec = _Spec()
    action_space = _Box(-np.ones(2), np.ones(2))
    observation_space = _Box(-np.inf * np.ones(6), np.inf * np.ones(6))
    horizon = 25                                   # (for mjrl_amd.samplers' env objects; mjrl's GymEnv reads spec.max_episode_steps)

    def __init__(self):
        self.rng = np.random.RandomState(0)

    def seed(self, s):
        self.rng = np.random.RandomState(s)

    set_seed = seed

    def _obs(self):
        return np.concatenate([self.p, self.v, self.g])

    def reset(self):
        self.p, self.v, self.g, self.t = self.rng.uniform(-1, 1, 2), np.zeros(2), self.rng.uniform(-1, 1, 2), 0
        return self._obs()

    def step(self, a):
        a = np.clip(np.asarray(a, np.float64), -1, 1)
        self.v = 0.9 * self.v + 0.1 * a
        self.p = self.p + 0.1 * self.v
        self.t += 1
        dist = float(np.linalg.norm(self.p - self.g))
        calls, forked, loaded = _libmjx_state()
        info = dict(pid=os.getpid(), mjx_device_calls=calls, mjx_forked_child=forked, mjx_loaded=loaded, goal_achieved=dist < 0.1)
        return self._obs(), -dist, dist < 0.02, info


def make_point_mass():
    """picklable factory (mjrl_amd.samplers' spawned workers import it by name)"""
    return PointMassGym()


def install_fake_gym():
    """bind ``gym.make`` / ``gym.Env`` of whatever module object the (possibly already imported) reference holds as ``gym``"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_loader
    if ref_loader.install() is None:
        return None
    import mjrl.utils.gym_env as ge
    gym = ge.gym
    if not isinstance(gym, types.ModuleType):
        raise RuntimeError("unexpected gym binding")

    def make(env_id, *a, **k):
        if env_id != ENV_ID:
            raise RuntimeError("no such env here: %r" % (env_id,))
        return PointMassGym()
    gym.make = make
    return ge


def worker_evidence(paths):
    """-> dict(pids, device_calls (max over all steps), forked (min over all steps), loaded) from the env_infos of `paths`"""
    pid = np.concatenate([np.atleast_1d(p["env_infos"]["pid"]) for p in paths])
    calls = np.concatenate([np.atleast_1d(p["env_infos"]["mjx_device_calls"]) for p in paths])
    forked = np.concatenate([np.atleast_1d(p["env_infos"]["mjx_forked_child"]) for p in paths])
    loaded = np.concatenate([np.atleast_1d(p["env_infos"]["mjx_loaded"]) for p in paths])
    return dict(pids=sorted(set(int(x) for x in pid)), device_calls=int(calls.max()), forked=int(forked.min()), loaded=int(loaded.max()))
