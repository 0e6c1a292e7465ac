"""Worker of tests/test_hip_end_to_end.py::test_two_processes_batch_runner: launched by torch.distributed.run with 2
processes that share cuda:0 (gloo collectives).  Every rank runs the slot-split BatchRunner and checks its results and
final genome against ig_step_batch on a private single-rank context."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from instagraal_amd import synth
    from instagraal_amd.multi_gpu import BatchRunner
    from instagraal_amd.sampler import sampler as hip_sampler

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    prob = synth.make_problem(*synth.CONFIGS["small"])

    def fresh():
        s = hip_sampler(**prob.sampler_kwargs(), device_id=0)
        s.set_param_simu(prob.params)
        s.eval_likelihood_init()
        return s

    np.random.seed(31)
    frags = np.resize(np.random.permutation(prob.n_frags), 250).astype(np.int32)
    ref = fresh()
    cands = ref.draw_candidates(frags, 5)
    want = ref.ctx.step_batch(frags, cands)
    want_state = ref.gpu_vect_frags.copy_from_gpu().soa17()
    s = fresh()
    got = BatchRunner(s.ctx, rank, world, dist=dist, width=12).run(frags, cands)
    assert got.tobytes() == want.tobytes(), "rank %d: results differ" % rank
    assert np.array_equal(s.gpu_vect_frags.copy_from_gpu().soa17(), want_state), "rank %d: genome differs" % rank
    dist.barrier()
    if rank == 0:
        print("TWO_RANK_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
