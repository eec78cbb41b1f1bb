"""
The N > 1 control flow of FlowTrainer on REAL RCCL, on the one GPU the test box has: a one-rank ``nccl`` process group with
NF_DP_FORCE_COLLECTIVE=1 (dist.GradBucket.collective) sends the trainer down the data-parallel path -- coalesced start-up broadcast,
graph A (zero + forward + NLL + backward), the flat bucket's all-reduce on RCCL, graph B (Adam); and, with one_graph, the all-reduce
captured INSIDE the step's hipGraph.  A one-rank all-reduce is the identity, so every mode must reproduce the plain single-process
trainer: what the test pins is that RCCL initialises on this stack (dmabuf IPC), that the collective runs on the
trainer's flat bucket, and that ProcessGroupNCCL's kernel can be captured and replayed between the hand-written kernels (tolerance: fp32 summation order of the atomic folds).
(world size 2 needs two GPUs: the gloo tests in test_dist_cpu.py cover the two-rank arithmetic.)  SURVEY.md section 8(e).
"""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = 'normalizing-flows-pytorch_amd'

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, mode, model, q):
    """mode: 'plain' (no process group), 'two' (graph A, eager RCCL all-reduce, graph B), 'one' (all-reduce captured)"""
    try:
        sys.path.insert(0, ROOT)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(0)
        nfdist = importlib.import_module(PKG + '.dist')
        train = importlib.import_module(PKG + '.train')
        models = importlib.import_module(PKG + '.models')
        if mode != 'plain':
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                              NF_DP_FORCE_COLLECTIVE='1')
            torch.distributed.init_process_group(backend='nccl', rank=0, world_size=1)
        torch.manual_seed(3)
        import numpy as np
        np.random.seed(3)

        class Cfg:
            layers = 4
        if model == 'glow_img':
            net, shape = models.Glow((3, 16, 16), 'image', Cfg), (32, 3, 16, 16)
        else:
            net, shape = models.RealNVP((2, ), None, Cfg), (256, 2)
        net = net.cuda()
        tr = train.FlowTrainer(net, lr=1e-3, graph=True, warmup=2, one_graph=(mode == 'one'))
        g = torch.Generator().manual_seed(11)
        y = torch.rand(shape, generator=g) if model == 'glow_img' else torch.randn(shape, generator=g)
        y = (y * 0.9 + 0.05).cuda() if model == 'glow_img' else y.cuda()
        losses = []
        for _ in range(6):
            _, loss = tr.train_on_batch(y)
            losses.append(float(loss))
        torch.cuda.synchronize()
        info = {'collective': tr.bucket.collective, 'graph': tr._g_fb is not None, 'two': tr._g_opt is not None,
                'whole': bool(getattr(tr, '_g_whole', False))}
        flat = tr.bucket.flat_params if tr.bucket.flat_params is not None else torch.cat([p.data.reshape(-1) for p in tr.bucket.params])
        q.put((mode, losses, flat.double().sum().item(), flat.double().abs().sum().item(), info))
        if mode != 'plain':
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
    except BaseException as e:                          # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((mode, 'error', traceback.format_exc(), 0.0, {}))
        raise


def _run(mode, model):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), mode, model, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res[1] != 'error', res[2]
    assert p.exitcode == 0
    return res


@pytest.mark.parametrize('model', ['realnvp_2d', 'glow_img'])
def test_trainer_on_a_one_rank_rccl_group_reproduces_the_single_process_trainer(model):
    plain = _run('plain', model)
    two = _run('two', model)
    one = _run('one', model)
    assert plain[4] == {'collective': False, 'graph': True, 'two': False, 'whole': True}
    assert two[4] == {'collective': True, 'graph': True, 'two': True, 'whole': False}
    assert one[4]['collective'] and one[4]['graph']
    # the captured all-reduce: one graph (whole) -- or the documented fallback to two graphs if ProcessGroupNCCL refuses capture
    assert one[4]['whole'] != one[4]['two']
    for r in (two, one):                                            # (float atomics in some weight-gradient folds: not bitwise)
        # step 1 runs from identical weights (rounding-level agreement); the later losses follow trajectories that two runs of the SAME
        # mode do not reproduce bit for bit either (order of the float atomics in the batch sums -> a ReLU decision on the other side of its
        # kink -> lr-sized differences after Adam; tools/probes/img_step2_dbg.py): the control flow is what is pinned here, at 2e-3
        assert r[1][0] == pytest.approx(plain[1][0], rel=1e-5), (r[0], r[1], plain[1])
        assert r[1] == pytest.approx(plain[1], rel=2e-3), (r[0], r[1], plain[1])        # losses of all six steps
        # parameters after them: the first Adam steps move every entry by ~ lr * sign(g), so entries whose gradient is rounding noise
        # around zero land lr apart between two runs of the SAME mode (float atomics): the absolute sum agrees to ~ steps * lr overall
        assert r[3] == pytest.approx(plain[3], rel=1e-3)
    assert one[4]['whole'], 'the RCCL all-reduce was not captured inside the step graph (fell back to two graphs)'
