"""Per-phase cycle stamps of one step of the persistent policy trainer (needs a build with -DMJX_PFIT_CLOCK)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.engine import UpdateEngine
from mjrl_amd._lib import check, ptr
import _synth as synth
n, m, hid, N, B, steps = 17, 6, (64, 64), 100000, 64, 8
rng = np.random.RandomState(0)
eng = UpdateEngine(n, m, hid)
th = torch.from_numpy(synth.perturbed_params(synth.init_params(n, m, hid))).cuda(); tho = th.clone()
obs, act, adv = [torch.from_numpy(rng.randn(*s).astype(np.float32)).cuda() for s in ((N, n), (N, m), (N,))]
idx = torch.from_numpy(rng.randint(0, N, size=(steps, B)).astype(np.int32)).cuda()
tr = torch.from_numpy(np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)).cuda()
am, av = torch.zeros_like(th), torch.zeros_like(th)
lt = torch.zeros(128, dtype=torch.float64).cuda()
for track in (1, 0):
    check(eng.lib.mjx_policy_minibatch_adam(eng.ctx, 2, ptr(obs), ptr(act), ptr(adv), ptr(idx), steps, B, ptr(th), ptr(tr), ptr(tho), ptr(tr), track,
                                            ptr(am), ptr(av), 0, 3e-4, 0.2, ptr(lt), eng.stream()))
    torch.cuda.synchronize()
    st = lt.cpu().numpy()[100:109]
    names = ["minibatch from registers + prefetch", "forward new", "forward old", "loss head", "wgrad3 + back3", "wgrad2 + back2", "wgrad1", "adam"]
    print("old_tracks_new =", track)
    for k, nm in enumerate(names):
        print("  %-38s %8d cycles" % (nm, st[k + 1] - st[k]))
    print("  step total %d cycles" % (st[8] - st[0]))
    fw = lt.cpu().numpy()[110:116]
    if fw[0] > 0:
        print("  forward detail (thread 0): layer1 %d + barrier %d | layer2 %d + barrier %d | layer3 %d + barrier %d" %
              (fw[0] - st[1], fw[1] - fw[0], fw[2] - fw[1], fw[3] - fw[2], fw[4] - fw[3], fw[5] - fw[4]))
