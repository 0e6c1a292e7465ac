"""GPU test: the collectives of instagraal_amd.multi_gpu through RCCL itself (backend "nccl") on the one GPU of this rig.

Two ranks may not share a device under RCCL, so the N > 1 protocol runs over gloo here (tests/test_hip_end_to_end.py,
tests/test_multi_gpu_protocol.py).  What those cannot show is that RCCL takes the library's own device buffers: the slot-major
record block of a batch (uint8 view of hipMalloc'ed memory, all_gather_into_tensor) and the int64 partial sums of a move
(all_reduce), ordered on the library's stream.  A process group of ONE rank does: same calls, same buffers, same stream -- the
collective degenerates to a copy onto itself, and the run must equal the plain one-GPU run byte for byte."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["IG_ROOT"])
from instagraal_amd import synth
from instagraal_amd.multi_gpu import BatchRunner, ShardedRunner
from instagraal_amd.sampler import sampler as hip_sampler

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert str(dist.get_backend()).lower() == "nccl"
prob = synth.make_problem(*synth.CONFIGS["small"])
N = 240


def fresh():
    s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
    s.set_param_simu(prob.params)
    s.eval_likelihood_init()
    np.random.seed(5)
    frags = np.resize(np.random.permutation(prob.n_frags), N).astype(np.int32)
    return s, frags


cols = ["o", "dist", "op_sampled", "id_f_sampled", "n_contigs"]
s, frags = fresh()
ref = s.step_sampler_batch(frags, 5)
ref_rows, ref_state = ref[cols].tobytes(), s.gpu_vect_frags.copy_from_gpu().soa17().tobytes()
s.free_gpu()

s, frags = fresh()
cands = s.draw_candidates(frags, 5)
r = BatchRunner(s.ctx, 0, 1, dist=dist, exchange_alone=True)
res = r.run(frags, cands)
torch.cuda.synchronize()
assert r._into_tensor is True and r.batches > 0
assert res[cols].tobytes() == ref_rows, "records differ behind the RCCL all-gather"
assert s.gpu_vect_frags.copy_from_gpu().soa17().tobytes() == ref_state
s.free_gpu()

s, frags = fresh()
cands = s.draw_candidates(frags, 5)
res = ShardedRunner(s.ctx, 0, 1, dist=dist).run(frags[:60], cands[:60])
torch.cuda.synchronize()
assert res[cols].tobytes() == ref[:60][cols].tobytes(), "records differ behind the RCCL all-reduce"
s.free_gpu()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK batches", r.batches)
"""


def test_the_collectives_run_through_rccl_on_the_librarys_buffers():
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    last = None
    for port in (29641, 29653, 29667):  # (a busy port: try the next)
        env = dict(os.environ, IG_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
        last = p.stdout[-1500:] + p.stderr[-3000:]
        if p.returncode == 0:
            assert "RCCL_ONE_RANK_OK" in p.stdout, last
            return
        if "address already in use" not in last.lower():
            break
    raise AssertionError(last)


def test_four_ranks_on_one_gpu_concurrently():
    """tools/fuzz_ranks.py's finding as a regression test: four ranks emulated as four contexts + four threads on the one GPU, on the
    shape that showed it (four contigs of 1 000 bins, nine neighbours, every column exact), 20 repetitions in one process.  Before the
    library's fills waited for themselves (memset_now: hipMemset runs on the null stream, the runners' torch streams are not ordered
    behind it) about one repetition in ten decided its first batch on a control block zeroed under it."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, FUZZ_WORLD="4", FUZZ_REPEAT="20", IG_SCREEN="0")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_ranks.py"), "20", "3090"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "20 cases, 0 bad" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
