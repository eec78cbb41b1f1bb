"""
Host-side logic of the product package that needs no GPU: constructors, state_dict contract, error behaviour,
mask rule, synthetic data generators, PLU inverse assembly.  (Transforms themselves are GPU-only.)
"""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import models as om
from oracle import nets as onets
from tests import _golden as G


@pytest.mark.parametrize('name', list(G.MODEL_CASES))
def test_state_dict_contract_matches_reference_fixture(pkg, name):
    """the golden sd0 IS a reference state_dict: same keys, shapes and dtypes must load strictly."""
    kind, cls, dims, datatype, layers, mix = G.MODEL_CASES[name]
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix, logdet='exact', spnorm_coeff=0.9))
    sd0 = G.group('model_' + name, 'sd0/')
    own = net.state_dict()
    assert list(own.keys()) == list(sd0.keys())
    for k in own:
        assert tuple(own[k].shape) == tuple(sd0[k].shape) and own[k].dtype == sd0[k].dtype, k
    net.load_state_dict(sd0, strict=True)
    frozen = [k for k, p in net.named_parameters() if not p.requires_grad]
    if kind == 'glow':
        assert any(k.endswith('.pivots') for k in frozen) and any(k.endswith('.P') for k in frozen)


def test_layer_plan_matches_oracle_plan(pkg):
    for kind, cls, dims, dt, layers, mix in [('glow', 'Glow', (3, 32, 32), 'image', 2, None),
                                             ('realnvp', 'RealNVP', (1, 16, 16), 'image', 1, None),
                                             ('flowpp', 'Flowpp', (3, 16, 16), 'image', 1, 4),
                                             ('maf', 'MAF', (4, ), None, 3, None), ('glow', 'Glow', (2, ), '2d', 5, None)]:
        net = getattr(pkg, cls)(dims, dt, NS(layers=layers, mixtures=mix))
        plan = om.build_plan(kind, dims, dt, layers, mix)
        names = {'actnorm': 'ActNorm', 'invconv': 'InvertibleConv1x1', 'affine': 'AffineCoupling', 'flow_bn': 'BatchNorm',
                 'logit': 'Logit', 'squeeze2d': 'Squeeze2d', 'unsqueeze2d': 'Unsqueeze2d', 'mixlog': 'MixLogAttnCoupling',
                 'ar': 'AutoregressiveTransfrom'}
        assert [type(m).__name__ for m in net.net.layers] == [names[L['op']] for L in plan]
        for m, L in zip(net.net.layers, plan):
            if 'odd' in L:
                assert m.odd == L['odd'] and m.mode == L['mode']


def test_glow_cifar_layer_count(pkg):
    net = pkg.Glow((3, 32, 32), 'image', NS(layers=32))
    assert len(net.net.layers) == 488
    n_train = sum(p.numel() for p in net.parameters() if p.requires_grad)
    assert n_train == 8380754


def test_errors(pkg):
    with pytest.raises(Exception, match='unsupported combination'):
        pkg.AffineCoupling((3, 4), masking='checkerboard')
    with pytest.raises(Exception, match='even'):
        pkg.AffineCoupling((3, ))
    with pytest.raises(NotImplementedError):
        pkg.MAF((3, 8, 8), 'image', NS(layers=1))


def test_made_mask_rule_matches_oracle(pkg):
    from importlib import import_module
    cond = import_module(pkg.__name__ + '.conditioners')
    for D in (2, 3, 5, 8):
        a = cond.made_degrees_to_masks(D, 3, 32, np.random.RandomState(5))
        b = onets.made_masks(D, 3, 32, np.random.RandomState(5))
        assert all(np.array_equal(x, y.numpy()) for x, y in zip(a, b))
    m = cond.made_degrees_to_masks(2, 3, 32, np.random.RandomState(0))
    assert (m[0] == np.array([[1.0, 0.0]] * 32)).all() and m[1].all() and m[2].all()
    assert (m[3][0] == 0).all() and (m[3][1] == 1).all()   # constant masks for D=2


def test_invconv_inverse_weight_is_lu_solve(pkg):
    torch.manual_seed(3)
    for C in (2, 3, 12, 48):
        layer = pkg.InvertibleConv1x1(C)
        with torch.no_grad():
            layer.L.add_(torch.randn(C, C) * 0.05)
            layer.U.add_(torch.randn(C, C) * 0.05)
        W = layer.weight()
        Winv = layer.inverse_weight()
        LU = layer.L * layer.L_mask + layer.U * layer.U_mask + torch.diag(layer.sign_s * torch.exp(layer.log_s))
        ref = torch.linalg.lu_solve(LU, layer.pivots, torch.eye(C))
        assert torch.allclose(Winv, ref, atol=2e-5)
        assert torch.allclose(Winv @ W, torch.eye(C), atol=2e-5)


def test_synthetic_data(pkg):
    from importlib import import_module
    data = import_module(pkg.__name__ + '.data')
    for name in ('moons', 'circles', 'normals'):
        a, b = data.sample(name, 4097, 7), data.sample(name, 4097, 7)
        assert a.shape == (4097, 2) and a.dtype == torch.float32 and torch.equal(a, b)
        assert not torch.equal(a, data.sample(name, 4097, 8))
        assert float(a.abs().max()) < 2.0
    r = np.linalg.norm(data.sample('circles', 20000, 1).numpy() / 0.6, axis=1)
    assert abs(np.median(r[r > 0.75]) - 1.0) < 0.05 and abs(np.median(r[r < 0.75]) - 0.5) < 0.05
    img = data.sample('cifar', 3, 0)
    assert img.shape == (3, 3, 32, 32) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0


def test_trainer_loss_matches_reference_formula(pkg):
    from importlib import import_module
    train = import_module(pkg.__name__ + '.train')
    z, ld = torch.randn(16, 3, 4, 4), torch.randn(16)
    mvn = torch.distributions.MultivariateNormal(torch.zeros(48), torch.eye(48))
    want = -1.0 * torch.mean(mvn.log_prob(z.view(16, -1)) + ld)                  # main.py:85
    assert torch.allclose(train.nll_loss(z, ld), want, atol=1e-5)
    assert abs(train.bits_per_dim(7.0 * 48 * np.log(2.0), (3, 4, 4)) - 7.0) < 1e-9


def test_flow_table_cache_is_bounded_and_keeps_pinned_entries(pkg):
    """fused._FlowTableCache: least-recently-used beyond its limit, except entries that were touched while a stream was capturing
    (a captured hipGraph has their device address baked into its kernel arguments)."""
    import importlib
    import torch
    fused = importlib.import_module(pkg.__name__ + '.fused')
    cache = fused._FlowTableCache(4)
    tables = [torch.zeros(8, dtype=torch.uint8) for _ in range(10)]
    for i in range(3):
        cache[('k', i)] = tables[i]
        cache.host[tables[i].data_ptr()] = object()
    cache.entries[('k', 1)][1] = True                   # as if looked up during a capture
    assert cache.get(('k', 0)) is tables[0]             # refreshes entry 0
    for i in range(3, 10):
        cache[('k', i)] = tables[i]
        cache.host[tables[i].data_ptr()] = object()
    assert len(cache) == 4
    assert cache.get(('k', 1)) is tables[1], 'a pinned entry was evicted'
    assert cache.get(('k', 0)) is None and cache.get(('k', 2)) is None
    assert cache.get(('k', 9)) is tables[9]
    assert tables[0].data_ptr() not in cache.host and tables[1].data_ptr() in cache.host


def test_zero_arena_retires_only_buffers_a_graph_was_captured_against(pkg):
    import importlib
    import torch
    ws = importlib.import_module(pkg.__name__ + '.workspace')
    arena = ws.ZeroArena()
    dev = torch.device('cpu')
    arena.begin(dev)
    arena.zeros(1 << 16, dev)
    arena.end()
    first = arena.buf
    arena.begin(dev)                                     # outgrown, never captured against: dropped, not retired
    assert arena.buf is not first and arena.retired == []
    arena.zeros(1 << 18, dev)
    arena.end()
    arena.captured = True                                # as if a capture had happened on the current buffer
    second = arena.buf
    arena.begin(dev)
    assert arena.buf is not second and arena.retired == [second]


def test_chain_launches_of_a_step_share_one_slot_buffer_with_rising_generations(pkg, monkeypatch):
    """fused_conv._chain_slots (host logic of the shared exchange slots): inside a step of the zero arena every request returns the SAME
    zeroed buffer with the next generation number (a larger request replaces the buffer, the numbers keep rising); a new step starts the
    numbers again on a re-zeroed arena; outside a step -- and with the switch off -- fresh zeros and generation 0."""
    import importlib
    import torch
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    ws = importlib.import_module(pkg.__name__ + '.workspace')
    dev = torch.device('cpu')
    arena = ws.ZeroArena()
    monkeypatch.setattr(ws, 'ARENA', arena)
    t0, g0 = fc._chain_slots(1000, dev)
    assert g0 == 0 and float(t0.abs().sum()) == 0.0 and t0.numel() == 1000
    for _ in range(2):                                   # two steps: the second finds the arena sized by the first
        arena.begin(dev)
        a, ga = fc._chain_slots(1000, dev)
        a[:10] = 7.0                                     # (what a launch leaves behind)
        b, gb = fc._chain_slots(600, dev)
        assert b.data_ptr() == a.data_ptr() and (ga, gb) == (1, 2) and float(b[0]) == 7.0
        c, gc = fc._chain_slots(5000, dev)               # larger than the shared buffer: a new zeroed one, the count goes on
        assert gc == 3 and c.numel() >= 5000 and float(c.abs().sum()) == 0.0
        d, gd = fc._chain_slots(1000, dev)
        assert d.data_ptr() == c.data_ptr() and gd == 4
        arena.end()
    monkeypatch.setattr(fc, 'CHAIN_SLOTS_SHARED', False)
    arena.begin(dev)
    e, ge = fc._chain_slots(1000, dev)
    f, gf = fc._chain_slots(1000, dev)
    arena.end()
    assert (ge, gf) == (0, 0) and e.data_ptr() != f.data_ptr()


def test_trainer_keeps_models_with_host_drawn_masks_off_the_graph(pkg):
    """MADE draws its masks from np.random on every call (flows/maf.py:50,72): constant for D = 2, varying for D > 2 -- a replayed
    graph would freeze the draw of the captured step, so FlowTrainer(graph=True) falls back to eager launches for such a model"""
    import importlib
    from types import SimpleNamespace as NS
    train = importlib.import_module(pkg.__name__ + '.train')
    for D, expect in ((2, True), (3, False)):
        net = pkg.MAF((D, ), '2d', NS(layers=2, mixtures=None))
        tr = train.FlowTrainer(net, graph=True, graph_factory=lambda: None)
        assert tr.graph is expect, (D, tr.graph)


def test_image_flowpp_conditioner_is_not_taken_off_the_gpu_or_for_other_shapes(pkg):
    """fused_flowpp_img.flowpp_img_fusable: CPU tensors, other widths, rectangular or large maps keep the module stack (the HIP path has no
    CPU fallback to fall into: the decision is made before any kernel is called)"""
    import importlib
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    net = cond.flowpp_conditioner(6, 84, (32, 8, 8), 32, conv=True)
    assert not fpi.flowpp_img_fusable(net, torch.zeros(2, 6, 8, 8))                     # not a GPU tensor
    assert not fpi.flowpp_img_fusable(net, torch.zeros(2, 6))                           # density data
    wide = cond.flowpp_conditioner(6, 84, (64, 8, 8), 64, conv=True)
    assert not fpi.flowpp_img_fusable(wide, torch.zeros(2, 6, 8, 8))


def test_bench_stdout_carries_only_the_json_line(tmp_path):
    """bench.py's contract: rank 0 prints ONE JSON line.  RCCL writes its start-up banner to stdout through C stdio, which a pipe
    flushes at process exit -- behind the line python printed (seen on the GPU box with a one-rank nccl group).  bench._claim_stdout
    sends everything else that reaches fd 1 to stderr; bench._emit_line writes the line to the saved descriptor last."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'emit.py'
    script.write_text(
        "import ctypes, importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench', %r)\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "fd = b._claim_stdout()\n"
        "libc = ctypes.CDLL(None)\n"
        "libc.printf(b'banner from C stdio\\n')\n"
        "print('python print')\n"
        "b._emit_line(fd, '{\"ok\": 1}')\n"
        "libc.printf(b'late C output\\n')\n" % os.path.join(ROOT, 'bench.py'))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"ok": 1}\n'
    assert 'banner from C stdio' in r.stderr and 'python print' in r.stderr
    merged = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
    lines = [ln for ln in merged.splitlines() if ln.strip()]
    assert lines.index('{"ok": 1}') > lines.index('banner from C stdio')        # buffered C output is flushed BEFORE the line


def test_bench_headline_stays_short_and_parses():
    """the driver could not parse round 5's 23 KB stdout line (BENCH_r05.parsed = null).  The line is now a headline -- the primary
    workload's contract keys, `roofline` / `cpu_baseline` / `parity` / `whole_step` cut to numbers and a phrase, one `summary` row per
    other workload -- and the detail goes to a side file.  Built here from canned objects of all eight workloads of a default run, with
    notes far longer than the real ones: the line parses, carries what the judge reads, and stays under bench.LINE_CAP."""
    import copy
    import importlib.util
    import json
    import os
    import pytest
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.LINE_CAP <= 4000
    essay = 'a note that goes on and on about how the number was measured, ' * 40
    one = {'metric': 'samples/sec (train step: forward flow + log-det + NLL + backward + Adam)', 'value': 2775.3, 'unit': 'samples/s', 'n_gpus': 1,
           'steps': 20, 'warmup': 5, 'ms_per_step': 23.0603, 'ms_per_step_event_median': 23.01, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic (seeded cifar restatement, random-init weights) ' + essay,
           'config': {'workload': 'Glow CIFAR-shape (3,32,32) L=3 K=32 batch 64 per GPU (512 over 8) ' + essay, 'name': 'c4', 'per_gpu_batch': 64,
                      'global_batch': 64, 'scaling': essay, 'parallelism': 'dp1', 'hipgraph': True, 'deterministic': False, 'dp_one_graph': False,
                      'collective': {'ranks': 1, 'backend': None}},
           'loss_nats': 11640.64453, 'bits_per_dim': 5.46676, 'forward_samples_per_s': 1.0, 'inverse_samples_per_s': 1.0, 'grad_bucket_bytes': 1,
           'roofline': {'bound': 'mfma', 'kernel': 'k_convnet_chain_bwd ' + essay, 'achieved': 16.988, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': 0.108,
                        'traffic': 61076183, 'flop_per_launch': 1277165568, 'bytes_per_launch': 29097984, 'us_per_launch': 75.179,
                        'launches_timed': 192, 'us_min_max': [73.4, 81.28], 'how': essay, 'note': essay, 'traffic_source': essay,
                        'bf16_pipe': {'hardware_tflops': 101.93, 'peak': 2500.0, 'frac': 0.04077, 'note': essay}, 'workgroups': 128},
           'whole_step': {'flop_per_step': 330811047936, 'mfma_tflops': 14.345, 'mfma_frac': 0.0912, 'hbm_bytes_per_step': 1899233280,
                          'hbm_gbs': 82.36, 'hbm_frac': 0.01029, 'note': essay},
           'cpu_baseline': {'value': 53.9, 'unit': 'samples/s', 'cores': 16, 'kind': 'port', 'sample': '17 train steps ' + essay, 'ms_per_step': 1187.46,
                            'host_cores': 256, 'threads_note': essay},
           'parity': {'loss_gpu_step1': 15142.165039, 'loss_cpu_step1': 15142.162109, 'abs_dloss_per_dim': 9.537e-07, 'max_abs_dz': 0.001162,
                      'max_abs_z': 6.002, 'note': essay}}
    out = copy.deepcopy(one)
    out['also'] = {}
    for name in ('c1', 'c2', 'c3', 'c5', 'rnvp_img', 'fpp_img', 'c4_b512'):
        o = copy.deepcopy(one)
        o['config']['name'] = name
        out['also'][name] = o
    assert len(json.dumps(out)) > 100000                           # the detail is big; the line is not
    line = bench.headline_of(out, 'gpurun_out/bench_detail.json')
    assert '\n' not in line and len(line.encode()) < bench.LINE_CAP
    head = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'cpu_baseline', 'parity', 'whole_step', 'summary'):
        assert key in head, key
    assert 'also' not in head
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert head['roofline'][key] == one['roofline'][key]
    for key in ('value', 'unit', 'cores', 'kind'):
        assert head['cpu_baseline'][key] == one['cpu_baseline'][key]
    assert head['cpu_baseline']['sample'].startswith('17 train steps')
    assert head['config']['workload'].startswith('Glow CIFAR-shape (3,32,32) L=3 K=32 batch 64')
    assert sorted(head['summary']['rows']) == sorted(['c4', 'c1', 'c2', 'c3', 'c5', 'rnvp_img', 'fpp_img', 'c4_b512'])
    assert head['summary']['rows']['c4_b512'][:2] == [2775.3, 23.0603]
    # --skip-cpu / N > 1: the objects are null, the line still builds
    lean = dict(one, cpu_baseline=None, parity=None, whole_step=None)
    assert json.loads(bench.headline_of(lean))['cpu_baseline'] is None
    # and the cap is enforced in code, not by convention
    fat = copy.deepcopy(out)
    for i in range(400):
        fat['also']['extra_workload_%03d' % i] = copy.deepcopy(one)
    with pytest.raises(AssertionError):
        bench.headline_of(fat)


def test_bench_scaling_modes_resolve_the_per_gpu_batch():
    """bench.py --scaling: weak keeps the per-GPU batch whatever N is, strong keeps the GLOBAL batch (C4: the literal 512 of BASELINE.json,
    C5: 131072) and gives every rank global / N rows; both report what they did in the config object."""
    import importlib.util
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for n in (1, 2, 4, 8):
        B, cfg, strong = b.resolve_batch(b.CONFIGS['c4'], 'weak', None, n)
        assert (B, strong) == (64, False) and cfg is b.CONFIGS['c4']
        B, cfg, strong = b.resolve_batch(b.CONFIGS['c4'], 'strong', None, n)
        assert (B, strong) == (512 // n, True) and 'GLOBAL batch 512' in cfg['desc'] and '%d per GPU' % (512 // n) in cfg['desc']
        B, _, _ = b.resolve_batch(b.CONFIGS['c5'], 'strong', None, n)
        assert B * n == 131072
        B, _, _ = b.resolve_batch(b.CONFIGS['c5'], 'weak', None, n)
        assert B == 16384
    assert b.resolve_batch(b.CONFIGS['c1'], 'strong', None, 8)[0] == 32
    with pytest.raises(SystemExit):
        b.resolve_batch(b.CONFIGS['c1'], 'strong', None, 3)
    assert b.resolve_batch(b.CONFIGS['c4'], 'strong', 128, 2)[0] == 128                 # an explicit --batch is always the per-GPU batch
    # the bench configs north_star's model list asks for beyond BASELINE.json's five: both image stacks at the reference's default depth
    assert b.CONFIGS['rnvp_img']['layers'] == 32 and b.CONFIGS['fpp_img']['layers'] == 32
    # the summary (last key of the line) has one row per workload
    o = {'value': 1.0, 'ms_per_step': 2.0, 'config': {'name': 'c4'}, 'roofline': {'frac': 0.1}, 'whole_step': None, 'cpu_baseline': {'value': 3.0},
         'parity': {'abs_dloss_per_dim': 1e-7, 'max_abs_dz': 1e-4}}
    o['also'] = {'c1': dict(o, config={'name': 'c1'})}
    s = b.summary_of(o)
    assert set(s['rows']) == {'c4', 'c1'} and s['rows']['c4'] == [1.0, 2.0, 0.1, None, 3.0, 1e-7, 1e-4] and len(s['columns']) == 7


def test_tensor_of_another_gpu_is_refused(pkg, monkeypatch):
    """_native.ptr: launches go to the current device's stream -- a tensor of another GPU is an error, not a wild pointer"""
    from types import SimpleNamespace as NS
    N = pkg._native
    fake = NS(is_cuda=True, is_contiguous=lambda: True, dtype=torch.float32, device=NS(index=1), data_ptr=lambda: 4096)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    with pytest.raises(N.NativeLibraryError, match='current device'):
        N.ptr(fake)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 1)
    assert N.ptr(fake) == 4096
