"""worker of test_gpu_multirank.py::test_nccl_group_without_in_library_rccl_falls_back_to_the_hook: one rank, an nccl process
group, libmjx's RCCL binding switched off (MJX_RCCL_DISABLE=1: as if librccl could not be loaded): the rank sums must come up on the hook transport (the same
C loops over torch.distributed's own RCCL group) and the update must carry the single-process bits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from mjrl_amd.engine import UpdateEngine
from tests._cases import NpgCase


def main(out):
    torch.cuda.set_device(0)
    c = NpgCase("npg_cfg2_small")
    tr = np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32)

    def update():
        eng = UpdateEngine(c.n, c.m, c.hidden)
        eng.set_policy(c.theta0, c.theta0, tr, tr)
        eng.set_batch(c.obs, c.act, c.adv_w)
        sa, kl = eng.npg_update(c.cg_iters, 1e-4, float(c.g["step"]), -3.0)
        th, kind = eng.theta_new.cpu().numpy().copy(), eng.comm_kind
        eng.close()
        return th, sa, kl, kind
    single = update()
    os.environ["MJX_COLLECTIVES_AT_WORLD1"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("MJX_TEST_PORT", "29977"), rank=0, world_size=1)
    grouped = update()
    dist.destroy_process_group()
    np.savez(out, same=np.array([np.array_equal(single[0], grouped[0]) and single[1:3] == grouped[1:3]]), kind=np.array([str(grouped[3])]),
             kind_single=np.array([str(single[3])]))


if __name__ == "__main__":
    main(sys.argv[1])
