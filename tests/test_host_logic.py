"""CPU tests (`-m "not gpu"`): the C ABI loads and exports what the header declares, the host-side mirrors of the
reference interfaces behave like the reference, and the product path refuses to run without a GPU / library."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from sid_lsg_amd._lib import HEADER, LIB_PATH, lib
    if not os.path.isfile(LIB_PATH):
        from sid_lsg_amd.csrc.build import build
        build(verbose=False)
    dll = lib.load()
    src = re.sub(r'/\*.*?\*/', ' ', open(HEADER).read(), flags=re.S)
    names = re.findall(r'\bint\s+(sidlsg_\w+)\s*\(', src)
    assert len(names) >= 30 and len(set(names)) == len(names)
    for n in names:
        assert hasattr(dll, n), f'{n} declared in include/sidlsg_hip.h but not exported'
    # host-only helpers are callable without a GPU
    assert lib.sidlsg_groupnorm_nchunks.raw(16, 4096, 320, 32) > 0
    assert lib.sidlsg_groupnorm_ws_floats.raw(16, 4096, 320, 33) < 0          # C % G != 0 -> rejected
    assert lib.sidlsg_layernorm_bwd_nblocks.raw(65536) > 0


def test_argument_validation_returns_einval_without_gpu():
    from sid_lsg_amd._lib import lib
    lib.load()
    # K not a multiple of 8, null pointers, bad stride: rejected before any launch
    assert lib.sidlsg_gemm_bf16.raw(None, 8, None, None, 8, None, None, 0, None, 0, 1, 4, 4, 8, 1.0, 0, None) == -22
    assert lib.sidlsg_conv3x3_bf16.raw(None, 8, None, None, 8, None, None, 0, None, 0, 1, 4, 4, 8, 8, 3, 0, 1.0, 0, None) == -22
    assert lib.sidlsg_adam_ema.raw(None, None, None, None, None, None, None, 0, 1, None) == -22


def test_product_ops_refuse_cpu_tensors():
    from sid_lsg_amd import ops
    from sid_lsg_amd.bias_act import bias_act
    with pytest.raises(RuntimeError):
        ops.silu(torch.zeros(8, 8, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):
        bias_act(torch.zeros(2, 8), torch.zeros(8), act='swish')            # impl='cuda' needs a GPU tensor
    # the explicit reference formulation stays available under its reference name
    x, b = torch.randn(2, 8, 3), torch.randn(8)
    y = bias_act(x, b, dim=1, act='swish', gain=1, impl='ref')
    torch.testing.assert_close(y, torch.nn.functional.silu(x + b[None, :, None]))


def test_get_plugin_builds_and_returns_a_module(tmp_path):
    """Loader seam B3 (torch_utils/custom_ops.py:46): get_plugin compiles the .hip sources with hipcc for gfx950, caches by
    module name, rebuilds only on a digest change, and returns a python module whose attributes are tensor-level
    functions (from <name>_binding.py) -- or the raw extern "C" symbols when there is no binding file."""
    import shutil
    from sid_lsg_amd import bias_act, custom_ops
    custom_ops.verbosity = 'none'
    assert bias_act._init()
    plug = bias_act._plugin
    assert plug.__name__ == 'bias_act_plugin' and callable(plug.bias_act) and hasattr(plug.dll, 'bias_act_plugin_launch')
    assert custom_ops.get_plugin('bias_act_plugin', sources=['ignored: cached by name']) is plug
    with pytest.raises(RuntimeError):                         # the tensor function refuses CPU tensors (no fallback)
        plug.bias_act(torch.zeros(4, 8), torch.zeros(8), torch.zeros(0), torch.zeros(0), torch.zeros(0), 0, 1, 9, 0.0, 1.0, -1.0)
    # a plugin without a binding file: raw C symbols, digest-keyed rebuild
    src = tmp_path / 'tiny_plugin.hip'
    src.write_text('#include <hip/hip_runtime.h>\nextern "C" int tiny_answer() { return 42; }\n')
    m = custom_ops.get_plugin('tiny_plugin', sources=[str(src)])
    assert m.dll.tiny_answer() == 42
    stamp = tmp_path / 'tiny_plugin.so.md5'
    first = stamp.read_text()
    custom_ops._cached_plugins.pop('tiny_plugin')
    src.write_text('#include <hip/hip_runtime.h>\nextern "C" int tiny_answer() { return 43; }\n')
    m2 = custom_ops.get_plugin('tiny_plugin2', sources=[str(shutil.copy(src, tmp_path / 'tiny_plugin2.hip'))])
    assert m2.dll.tiny_answer() == 43 and (tmp_path / 'tiny_plugin2.so.md5').read_text() != first


def test_unet_structure_matches_diffusers_contract():
    from oracle.unet_ref import CONFIGS as RC, UNet2DConditionRef
    from sid_lsg_amd.unet import CONFIGS, HipUNet2DCondition
    for name, total in (('sd15', 859520964), ('sd21-base', 865910724)):
        hip = HipUNet2DCondition(CONFIGS[name])            # meta parameters: no memory
        with torch.device('meta'):
            ref = UNet2DConditionRef(RC[name])
        a = {k: tuple(v.shape) for k, v in hip.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert a == b
        assert sum(p.numel() for p in hip.parameters()) == total
    assert 'down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight' in a
    assert 'up_blocks.3.resnets.2.conv_shortcut.weight' in a and a['conv_in.weight'] == (320, 4, 3, 3)


def test_infinite_sampler_and_prompt_dataset(golden_dir, tmp_path):
    from sid_lsg_amd.data import InfiniteSampler, PromptDataset, prompt_batches
    g = np.load(os.path.join(golden_dir, 'sampler.npz'))
    for key in g.files:                      # order captured from the reference's torch_utils/misc.py::InfiniteSampler
        n, r, w, s = (int(x[1:]) for x in key.split('_'))
        it = iter(InfiniteSampler(list(range(n)), rank=r, num_replicas=w, seed=s))
        assert [next(it) for _ in range(64)] == g[key].tolist()
    (tmp_path / 'aesthetics_625_plus.txt').write_text('a cat\na dog\n\n')
    ds = PromptDataset(str(tmp_path), resolution=512)
    assert len(ds) == 3 and ds[1][1] == 'a dog' and ds[0][0].shape == (1, 4, 4) and ds.name == 'aesthetics'
    batches = prompt_batches(ds, InfiniteSampler(ds, seed=1), 2)
    assert len(next(batches)) == 2


def test_dnnlib_seam():
    from sid_lsg_amd.dnnlib_util import EasyDict, construct_class_by_name, get_obj_by_name
    d = EasyDict(a=1)
    d.b = 2
    assert d['b'] == 2 and d.a == 1
    assert get_obj_by_name('sid_lsg_amd.optim.FusedAdamEMA').__name__ == 'FusedAdamEMA'
    sched = construct_class_by_name(class_name='sid_lsg_amd.scheduler.DDPMScheduler')
    assert abs(float(sched.alphas_cumprod[625]) - 0.13776892) < 2e-7
    with pytest.raises(ImportError):
        construct_class_by_name(class_name='sid_lsg_amd.does_not.Exist')


def test_scheduler_matches_oracle():
    from oracle.scheduler_ref import DDPMSchedulerRef
    from sid_lsg_amd.scheduler import DDPMScheduler
    a, b = DDPMScheduler(), DDPMSchedulerRef()
    g = torch.Generator().manual_seed(0)
    x0, n = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g)
    t = torch.tensor([20, 625, 979])
    torch.testing.assert_close(a.add_noise(x0, n, t), b.add_noise(x0, n, t))
    xt = a.add_noise(x0, n, t)
    ref = torch.stack([b.step(e, tt, z).pred_original_sample for e, tt, z in zip(n, t, xt)])
    torch.testing.assert_close(a.step(n, t, xt).pred_original_sample, ref)          # vectorised == per-sample loop (sid_sd_util.py:270)
    assert a.config.prediction_type == 'epsilon'


def test_glue_refuses_foreign_networks():
    """No generic / CPU branch in the product glue: anything that is not a HipUNet2DCondition is rejected (the glue's
    arithmetic is pinned to the reference through oracle/sid_ref.py + tests/golden, and on the GPU through
    test_gpu_unet.py::test_glue_matches_reference_golden)."""
    from oracle import fixtures
    from sid_lsg_amd.scheduler import DDPMScheduler
    from sid_lsg_amd.sd_util import sid_sd_denoise, sid_sd_sampler
    unet, _, _, te, tok = fixtures.factory('tiny')
    z = torch.zeros(1, 4, 8, 8)
    t = torch.full((1,), 625, dtype=torch.long)
    with pytest.raises(TypeError):
        sid_sd_sampler(unet, z, ['a'], t, DDPMScheduler(), te, tok, 64)
    with pytest.raises(TypeError):
        sid_sd_denoise(unet, z, z, ['a'], t, DDPMScheduler(), te, tok, 64)


def test_text_encoder_matches_transformers():
    transformers = pytest.importorskip('transformers')
    from oracle.text_ref import CLIPTextRef
    from sid_lsg_amd.text import CLIPTextModel, HashTokenizer
    cfg = transformers.CLIPTextConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_hidden_layers=2,
                                      max_position_embeddings=77, vocab_size=49408)
    torch.manual_seed(0)
    hf = transformers.CLIPTextModel(cfg).eval()
    ours = CLIPTextModel(hidden=64, layers=2, heads=4, dff=128).eval()
    # SD checkpoints / transformers==4.40.1 (the reference's pin) prefix the keys with `text_model.`; transformers 5.x dropped it
    hf_sd = {(k if k.startswith('text_model.') else 'text_model.' + k): v for k, v in hf.state_dict().items()}
    missing, unexpected = ours.load_state_dict(hf_sd, strict=False)     # same key names as transformers
    assert not missing, missing
    ids = HashTokenizer()(['a photo of a cat', '']).input_ids
    with torch.no_grad():
        ref = hf(ids)[0]
        got = ours(ids)[0]
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)
    # the oracle restatement agrees as well (it is what the golden generator used)
    o = CLIPTextRef(64, 2, 4, 128).eval()
    sd = hf_sd
    with torch.no_grad():
        o.token_embedding.weight.copy_(sd['text_model.embeddings.token_embedding.weight'])
        o.position_embedding.weight.copy_(sd['text_model.embeddings.position_embedding.weight'])
        for i, l in enumerate(o.layers):
            p = f'text_model.encoder.layers.{i}.'
            for a, b in (('q_proj', 'self_attn.q_proj'), ('k_proj', 'self_attn.k_proj'), ('v_proj', 'self_attn.v_proj'),
                         ('out_proj', 'self_attn.out_proj'), ('fc1', 'mlp.fc1'), ('fc2', 'mlp.fc2'),
                         ('layer_norm1', 'layer_norm1'), ('layer_norm2', 'layer_norm2')):
                getattr(l, a).weight.copy_(sd[p + b + '.weight']); getattr(l, a).bias.copy_(sd[p + b + '.bias'])
        o.final_layer_norm.weight.copy_(sd['text_model.final_layer_norm.weight'])
        o.final_layer_norm.bias.copy_(sd['text_model.final_layer_norm.bias'])
        torch.testing.assert_close(o(ids)[0], ref, rtol=1e-4, atol=1e-5)


def test_bpe_tokenizer_roundtrip(tmp_path):
    import json
    from sid_lsg_amd.text import CLIPBPETokenizer
    vocab = {'a</w>': 0, 'c': 1, 'a': 2, 't</w>': 3, 'ca': 4, 'cat</w>': 5, 'at</w>': 6}
    (tmp_path / 'vocab.json').write_text(json.dumps(vocab))
    (tmp_path / 'merges.txt').write_text('#version\nc a\nca t</w>\n')
    tok = CLIPBPETokenizer.from_files(str(tmp_path / 'vocab.json'), str(tmp_path / 'merges.txt'), model_max_length=8)
    ids = tok(['a cat']).input_ids[0].tolist()
    assert ids[:4] == [49406, 0, 5, 49407] and ids[4:] == [49407] * 4


def test_flat_grad_reducer_world2_gloo(tmp_path):
    """N>1 path on CPU: two gloo ranks all-reduce a flat gradient buffer in buckets; sum + 1/world == DDP mean."""
    script = tmp_path / 'w.py'
    script.write_text(f'''
import os, sys, torch
sys.path.insert(0, {ROOT!r})
from sid_lsg_amd import distributed as dist
dist.init(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = torch.arange(10007, dtype=torch.float32) * (rank + 1)
red = dist.FlatGradReducer(nbuckets=3)
red.start(g); red.wait()
expect = torch.arange(10007, dtype=torch.float32) * 3
assert torch.equal(g, expect), (g[:4], expect[:4])
assert torch.allclose(g / world, torch.arange(10007, dtype=torch.float32) * 1.5)
# segment-wise exchange (overlap of the generator's exchange with its own backward): ranges in completion order,
# several in flight, one wait; plus the autograd marker that triggers them
g2 = torch.arange(10007, dtype=torch.float32) * (rank + 1)
fired = []
x = torch.ones(4, requires_grad=True)
from sid_lsg_amd import ops
def cb(k, segs=[(7000, 10007), (4000, 7000)]):
    fired.append(k); red.start_range(g2, *segs[k], max_elems=1024)
h = ops.grad_ready_marker(x * 2.0, lambda: cb(1))
h = ops.grad_ready_marker(h * 3.0, lambda: cb(0))
(h * 5.0).sum().backward()
assert fired == [0, 1] and torch.equal(x.grad, torch.full((4,), 30.0))
red.start_range(g2, 0, 4000); red.wait()
assert torch.equal(g2, expect)
# opt-in bf16 exchange: each rank's contribution is rounded to bf16, the sum comes back into the fp32 buffer
g3 = (torch.arange(10007, dtype=torch.float32) * 0.37 + 1.0) * (rank + 1)
red16 = dist.FlatGradReducer(nbuckets=3, exchange_dtype=torch.bfloat16)
red16.start(g3); red16.wait()
want = (torch.arange(10007, dtype=torch.float32) * 0.37 + 1.0) * 3
assert g3.dtype == torch.float32 and float(((g3 - want).abs() / want).max()) < 2.0 ** -7, float(((g3 - want).abs() / want).max())
assert not torch.equal(g3, want)          # it IS a lossy exchange: off by default
assert dist.FlatGradReducer().exchange_dtype == torch.float32
dist.print0("REDUCER_OK", world)
''')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                          '127.0.0.1', '--master-port', '29631', str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'REDUCER_OK 2' in out.stdout


def test_cli_dry_run_builds_reference_config(tmp_path):
    """sid_train.py (boundary B5): option surface + derived config of the reference CLI, no GPU needed for --dry-run."""
    from click.testing import CliRunner
    import sid_train
    prompts = tmp_path / 'aesthetics_6_plus.txt'
    prompts.write_text('a red cube\na blue sphere\n')
    res = CliRunner().invoke(sid_train.main, [
        '--outdir', str(tmp_path / 'runs'), '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny', '--seed', '3',
        '--batch', '8', '--batch-gpu', '2', '--duration', '0.01', '--ema', '0.05', '--cfg_train_fake', '1.5',
        '--cfg_eval_fake', '1.5', '--cfg_eval_real', '1.5', '--optimizer', 'adamw', '--fp16', '1', '--dry-run'])
    assert res.exit_code == 0, res.output
    assert 'Dry run; exiting.' in res.output
    o = sid_train.EasyDict(dict(
        outdir='x', data=None, data_stat=None, data_prompt_text=str(tmp_path), duration=0.01, batch=8, batch_gpu=2, ema=0.05,
        xflip=0.0, bench=True, cache=True, workers=1, desc=None, nosubdir=False, tick=2, snap=50, dump=100, seed=3, transfer=None,
        resume=None, dry_run=True, metrics=None, sd_model='random:tiny', resolution=512, init_timestep=625, fp16=True, ls=1, lsg=1,
        alpha=1, tmax=980, tmin=20, lr=1e-6, glr=2e-6, train_mode=True, network_pkl=None, cfg_train_fake=1.5, cfg_eval_fake=1.5,
        cfg_eval_real=1.5, metric_pt_path=None, metric_clip_path=None, metric_open_clip_path=None, enable_xformers=True,
        gradient_checkpointing=False, optimizer='adamw', num_steps=1, fake_score_use_lora=False))
    c = sid_train.build_config(o)
    assert c.total_kimg == 10 and c.ema_halflife_kimg == 50 and c.batch_size == 8 and c.batch_gpu == 2
    assert c.g_optimizer_kwargs == dict(class_name='torch.optim.AdamW', lr=2e-6, betas=[0.0, 0.999], eps=1e-6, weight_decay=0.01)
    assert c.fake_score_optimizer_kwargs['lr'] == 1e-6
    assert c.dataset_prompt_text_kwargs['class_name'] == 'sid_lsg_amd.data.PromptDataset'
    assert c.network_kwargs == dict(use_fp16=True, compute_dtype='bf16', teacher_weights='bf16')
    # errors the reference raises as ClickException
    bad = CliRunner().invoke(sid_train.main, ['--outdir', 'x', '--data_prompt_text', str(tmp_path), '--resume', 'nope.pt', '--seed', '0', '-n'])
    assert bad.exit_code != 0 and 'training-state' in bad.output
    # --metrics with an empty --data directory is refused at start-up (the caption set is only constructed at the first metrics tick)
    det, stat, empty = tmp_path / 'det.pt', tmp_path / 'stat.npz', tmp_path / 'no_captions'
    det.write_bytes(b'x'); stat.write_bytes(b'x'); empty.mkdir()
    bad = CliRunner().invoke(sid_train.main, ['--outdir', 'x', '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny', '--metrics', 'fid30k_full',
                                              '--metric_pt_path', str(det), '--data_stat', str(stat), '--data', str(empty), '-n'])
    assert bad.exit_code != 0 and 'no captions' in bad.output, bad.output


def test_generate_cli_helpers():
    import generate_onestep as g
    assert g.parse_int_list('1,2,5-8') == [1, 2, 5, 6, 7, 8]
    a = g.StackedRandomGenerator('cpu', [7, 8]).randn([2, 4, 3, 3])
    b = g.StackedRandomGenerator('cpu', [8]).randn([1, 4, 3, 3])
    assert torch.equal(a[1], b[0])          # a sample depends on its own seed only, not on the batch


@pytest.mark.parametrize('golden', ['loop_k15_a1', 'loop_k1_a12', 'loop_k45_a1'])
def test_prompt_stream_matches_reference_rng_order(golden_dir, tmp_path, golden):
    """the PRODUCT loop's input stream (prompt order, dropout flags, z, noise, t) equals, draw
    for draw, the stream of oracle.sid_ref.training_loop_ref -- the restatement that reproduces the loss curves the
    UNMODIFIED reference training_loop produced for tests/golden/loop_*.npz (tests/test_oracle_pinned.py), including the
    base-seed draw of iter(DataLoader) (sid_training_loop.py:275).  Everything on the CPU generator (rng_device='cpu')."""
    from oracle import fixtures, sid_ref
    from sid_lsg_amd.data import PromptDataset
    from sid_lsg_amd.training_loop import PromptStream
    g = np.load(os.path.join(golden_dir, golden + '.npz'))
    prompts = [str(p) for p in g['prompts']]
    kw = dict(iterations=int(g['kw_iterations']), batch_size=int(g['kw_batch_size']), batch_gpu=int(g['kw_batch_gpu']),
              seed=int(g['kw_seed']), kappa=tuple(float(k) for k in g['kw_kappa']), resolution=int(g['kw_resolution']))
    # --- oracle stream: record the inputs of every iteration
    recorded = []
    orig_iter = sid_ref.sid_iteration_ref
    try:   # replace the compute by a recorder: the stream must not depend on it
        sid_ref.sid_iteration_ref = lambda nets, st, sched, inputs, hp: (recorded.append(inputs), dict(loss_fake=0.0, loss_G=0.0))[1]
        sid_ref.training_loop_ref(lambda: fixtures.factory(str(g['cfg'])), prompts, alpha=float(g['kw_alpha']), **kw)
    finally:
        sid_ref.sid_iteration_ref = orig_iter
    # --- product stream
    pdir = tmp_path / 'p'
    pdir.mkdir()
    (pdir / 'aesthetics_6_plus.txt').write_text('\n'.join(prompts) + '\n')
    ds = PromptDataset(str(pdir), resolution=kw['resolution'])
    rounds = kw['batch_size'] // kw['batch_gpu']
    _, _, _, te, tok = fixtures.factory(str(g['cfg']))       # (module construction consumes RNG: before the stream is seeded)
    stream = PromptStream(ds, seed=kw['seed'], rank=0, world=1, batch_gpu=kw['batch_gpu'], lat=kw['resolution'] // 8, tmin=20,
                          tmax=980, device='cpu', rng_device='cpu')
    for _ in range(16):
        stream.next_prompts()
    use_dropout = kw['kappa'][0] != 1 or kw['kappa'][1] != 1

    def embed(ps):
        ids = tok(ps, padding='max_length', max_length=tok.model_max_length, truncation=True, return_tensors='pt').input_ids
        with torch.no_grad():
            return te(ids)[0]
    ndrop = 0
    for it in range(kw['iterations']):
        for ph, drop in (('A', use_dropout), ('B', False)):
            for r in range(rounds):
                ps, z, noise, t = stream.round(drop)
                ref = recorded[it][ph][r]
                assert torch.equal(z, ref['z']) and torch.equal(noise, ref['noise']) and torch.equal(t, ref['t']), (it, ph, r)
                assert torch.equal(embed(ps), ref['cond']), f'prompts / dropout flags differ at iteration {it} phase {ph} round {r}'
                ndrop += sum(p == '' for p in ps)
    print(golden, 'dropped prompts in the stream:', ndrop)


def test_frechet_distance_and_feature_stats():
    """sid_lsg_amd.metrics: the Frechet distance against the reference's formula
    (metrics/sid_fid_and_clip.py:65-67, `scipy.linalg.sqrtm`), incl. rank-deficient covariances (fewer samples than
    features); FeatureStats against numpy (sid_metric_utils.py:135-176) incl. the max_items cut; the CLIP score."""
    import scipy.linalg
    from sid_lsg_amd import metrics
    rng = np.random.default_rng(0)
    for n, f in ((500, 64), (40, 64), (300, 96)):
        a = rng.normal(size=(n, f)) @ rng.normal(size=(f, f)) * 0.3 + rng.normal(size=f)
        b = rng.normal(size=(n + 7, f)) @ rng.normal(size=(f, f)) * 0.25 - 0.1
        mu_a, mu_b = a.mean(0), b.mean(0)
        s_a, s_b = np.cov(a, rowvar=False, bias=True), np.cov(b, rowvar=False, bias=True)
        m = np.square(mu_a - mu_b).sum()
        sq, _ = scipy.linalg.sqrtm(np.dot(s_a, s_b), disp=False)
        ref = float(np.real(m + np.trace(s_a + s_b - sq * 2)))
        got = metrics.frechet_distance(mu_a, s_a, mu_b, s_b)
        assert abs(got - ref) <= 1e-6 * max(1.0, abs(ref)), (n, f, got, ref)
        assert abs(metrics.frechet_distance(mu_a, s_a, mu_a, s_a)) < 1e-6 * np.trace(s_a)
    x = rng.normal(size=(257, 33)).astype(np.float32)
    st = metrics.FeatureStats(capture_mean_cov=True, capture_all=True, max_items=200)
    for i in range(0, 257, 50):
        st.append(torch.from_numpy(x[i:i + 50]))
    assert st.is_full() and st.num_items == 200 and st.get_all().shape == (200, 33)
    mu, sigma = st.get_mean_cov()
    x64 = x[:200].astype(np.float64)
    assert np.allclose(mu, x64.mean(0), atol=1e-12) and np.allclose(sigma, x64.T @ x64 / 200 - np.outer(x64.mean(0), x64.mean(0)), atol=1e-10)
    img, txt = rng.normal(size=(10, 8)), rng.normal(size=(10, 8))
    assert abs(metrics.clip_score_from_features(np.concatenate([img, txt], 1)) - (img * txt).sum(-1).mean()) < 1e-9


def test_metric_registry_and_report(tmp_path):
    """calc_metric / report_metric with the reference's metric names and jsonl file names (sid_metric_main.py:25-123), on a
    CPU stand-in generator and detector; two ranks must merge their statistics to the one-rank result."""
    from sid_lsg_amd import metrics
    assert set(metrics.list_valid_metrics()) >= {'fid30k_full', 'fid_clip_30k_full', 'fid_test', 'fid_clip_test'}
    torch.manual_seed(0)
    proj = torch.randn(3 * 16 * 16, 24) * 0.01
    seen = []

    def G(latents, contexts, init_timesteps):
        assert len(contexts) == latents.shape[0] and init_timesteps.shape[0] == latents.shape[0]
        return torch.tanh(torch.nn.functional.interpolate(latents[:, :3], scale_factor=8.0))       # [-1, 1] images at 8x the latent size

    def detector(img, return_features=True):
        assert img.dtype == torch.uint8 and img.shape[1:] == (3, 16, 16)
        f = img.float().flatten(1) @ proj
        seen.append(f)
        return f

    def oc(img, texts, div255):
        f = torch.nn.functional.normalize(img.float().flatten(1)[:, :8], dim=1)
        return torch.cat([f, f], 1)                                                                # cosine 1 with "its text"
    real = (np.zeros(24), np.eye(24))
    res = metrics.calc_metric('fid_clip_test', G=G, prompts=['a', 'b', 'c'], resolution=16, detector=detector, real_stats=real,
                              open_clip_detector=oc, device='cpu', num_test=12, detector_size=16, batch_gen=5)
    feats = torch.cat(seen).double().numpy()
    assert feats.shape == (12, 24)
    ref = metrics.frechet_distance(feats.mean(0), np.cov(feats, rowvar=False, bias=True), *real)
    assert abs(res.results.fid30k_full - ref) < 1e-8 * max(1.0, abs(ref)) and abs(res.results.open_clipscore_30k - 1.0) < 1e-6
    assert np.isnan(res.results.clipscore30k) and res.metric == 'fid_clip_test' and res.num_gpus == 1
    run = tmp_path / 'run'
    run.mkdir()
    metrics.report_metric(res, run_dir=str(run), snapshot_pkl=str(run / 'network-snapshot-1.000000-000010.pkl'), alpha=1.0)
    line = json.loads((run / 'metric-fid_clip_test-alpha-1.000000.jsonl').read_text())
    assert line['snapshot_pkl'] == 'network-snapshot-1.000000-000010.pkl' and line['results']['fid30k_full'] == res.results.fid30k_full
    with pytest.raises(FileNotFoundError):
        metrics.load_detector('/nonexistent/inception-2015-12-05.pt', 'cpu')
    # two gloo ranks: rank-strided samples, statistics merged with one all_reduce
    script = tmp_path / 'm.py'
    script.write_text(f'''
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from sid_lsg_amd import distributed as dist, metrics
dist.init(backend="gloo")
torch.manual_seed(0)
proj = torch.randn(3 * 16 * 16, 24) * 0.01
G = lambda latents, contexts, init_timesteps: torch.tanh(torch.nn.functional.interpolate(latents[:, :3], scale_factor=8.0))
det = lambda img, return_features=True: img.float().flatten(1) @ proj
o = metrics.MetricOptions(G=G, prompts=["a", "b", "c"], resolution=16, detector=det, real_stats=(np.zeros(24), np.eye(24)), device="cpu",
                          detector_size=16, batch_gen=3)
stats, _, _ = metrics.generator_feature_stats(o, 10)
assert stats.num_items == 10, stats.num_items
mu, sigma = stats.get_mean_cov()
assert np.isfinite(mu).all() and np.allclose(sigma, sigma.T)
dist.print0("METRICS_OK", dist.get_world_size(), float(metrics.frechet_distance(mu, sigma, np.zeros(24), np.eye(24))))
''')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                          '127.0.0.1', '--master-port', '29633', str(script)], capture_output=True, text=True,
                         env=dict(os.environ, MASTER_ADDR='127.0.0.1'), timeout=300)
    assert out.returncode == 0 and 'METRICS_OK 2' in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]


def replay_metrics_golden(golden_dir, device):
    """Our evaluation pipeline (sid_lsg_amd.metrics) on `device` against tests/golden/metrics_ref.npz -- vectors produced by the
    REFERENCE's own `compute_feature_stats_for_generator` / `FeatureStats` / `resize_images_in_tensor` (metrics/sid_metric_utils.py)
    and the Frechet formula of metrics/sid_fid_and_clip.py:65-67, on the stand-in generator / detector of oracle/fixtures.py."""
    from oracle import fixtures
    from sid_lsg_amd import metrics
    g = np.load(os.path.join(golden_dir, 'metrics_ref.npz'))
    assert [str(c) for c in g['captions']] == fixtures.METRIC_CAPTIONS
    R, N = int(g['resolution']), int(g['num_items'])
    det = fixtures.StandInDetector().to(device)
    seen = dict(contexts=[], resized=[])

    def G(latents, contexts, init_timesteps):
        assert latents.shape[1:] == (4, R // 8, R // 8) and latents.device.type == torch.device(device).type
        seen['contexts'].extend(contexts)
        return fixtures.text_images(contexts, R).to(device)

    def detector(img, return_features=True):
        seen['resized'].append(img.clone())
        return det(img)
    res = metrics.calc_metric('fid_test', G=G, dataset_kwargs=dict(class_name='oracle.fixtures.CaptionSet', resolution=R), resolution=R,
                              init_timestep=625, detector=detector, real_stats=(g['mu_real'], g['sigma_real']), device=device, num_test=N)
    # 1. prompt order = the reference's InfiniteSampler(seed = 0) stream through its DataLoader
    assert seen['contexts'] == [str(c) for c in g['contexts']]
    # 2. uint8 conversion + PIL LANCZOS resize to 256 x 256: bit for bit
    resized = torch.cat(seen['resized'])
    assert resized.shape == (N, 3, 256, 256) and resized.dtype == torch.uint8
    assert np.array_equal(resized[0].cpu().numpy(), g['resized_first'])
    assert np.array_equal(resized.to(torch.float64).sum(dim=(2, 3)).cpu().numpy(), g['resized_sums'])
    # 3. features -> FeatureStats mean / covariance -> Frechet distance
    o = metrics.MetricOptions(G=G, dataset_kwargs=dict(class_name='oracle.fixtures.CaptionSet', resolution=R), resolution=R, detector=detector,
                              real_stats=(g['mu_real'], g['sigma_real']), device=device)
    stats, _, _ = metrics.generator_feature_stats(o, N)
    mu, sigma = stats.get_mean_cov()
    assert stats.num_items == int(g['stats_num_items'])
    np.testing.assert_allclose(mu, g['mu_gen'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sigma, g['sigma_gen'], rtol=1e-4, atol=1e-8)
    fid = res.results.fid30k_full
    print(f'[{device}] FID {fid:.9f} vs the reference path {float(g["fid"]):.9f}')
    assert abs(fid - float(g['fid'])) < 1e-5 * abs(float(g['fid']))
    return fid


def test_metrics_pipeline_matches_reference_golden_cpu(golden_dir):
    replay_metrics_golden(golden_dir, 'cpu')


def test_bench_line_is_compact():
    """The driver keeps the last ~8 KB of bench.py's stdout (round 4's 22 KB line lost its head: `parsed: null`).  The ONE stdout
    line is built by bench.compact_line from the full result dict: < 4 KB even with every family present and paragraph-long
    strings in the full objects, and it carries the contract's keys + `roofline` + `cpu_baseline`."""
    import bench
    fam = {'bound': 'mfma', 'kernel': 'k' * 300, 'achieved': 316.92153742307295, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.12676861496922917,
           'timing': 't' * 400, 'frac_of_per_call_roofline': 0.1630670197494531, 'algorithmic_GBps': 1073.8, 'frac_hbm': 0.134, 'traffic': None,
           'launches': 3935, 'launches_total': 27540, 'kernels_timed': 4276, 'avg_launch_ms': 0.07589743257946367, 'est_ms_per_step': 104.51076466192148,
           'algorithmic_tflop_per_launch': 0.024, 'isolated': {'achieved': 585.5, 'frac': 0.2342061989362917, 'avg_launch_ms': 0.0409, 'launches': 591, 'what': 'w' * 200}}
    full = {'metric': 'distillation images/sec (512^2, SD1.5, kappa=1.5)', 'value': 38.43796376494744, 'unit': 'images/s', 'n_gpus': 8, 'steps': 20, 'warmup': 5,
            'ms_per_step': 208.12757015228271, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'w' * 400, 'global_batch': 64, 'parallelism': 'dp8', 'teacher_weights': 'bf16'}, 'step_tflops': 555.77, 'step_mfma_frac': 0.2223,
            'graph': False, 'grouped_frozen_pass': False, 'host_enqueue_ms_per_step': None, 'loss_fake': 15833.16, 'loss_G': 3399.38, 'peak_mem_gb': 69.2,
            'loss_check': 'ok', 'loss_check_detail': [{'x': 'y' * 500}] * 2,
            'comm': {'exposed_ms_per_step': {'fake_score': 1.234567, 'G': 7.654321}, 'comm_exposed_ms': 8.9, 'messages_per_step': 8.0, 'bytes_per_step': 6.9e9,
                     'allreduce_ms_per_step': 40.1, 'allreduce_algbw_GBps': 171.123456, 'allreduce_busbw_GBps': 299.4, 'backend': 'nccl', 'note': 'n' * 300},
            'teacher_pass': {'what': 'x' * 100, 'ms': 18.3, 'tflops': 700.0, 'mfma_frac': 0.28},
            'frozen_pair_pass': {'what': 'x' * 200, 'ms': 34.5, 'tflops': 740.0, 'mfma_frac': 0.296, 'vs_two_separate_passes': 1.06},
            'roofline': dict(fam, family='gemm', why='y' * 100),
            'cpu_baseline': {'value': 0.0256, 'unit': 'images/s', 'cores': 32, 'kind': 'port', 'sample': 's' * 400, 'detail': 'd' * 400}}
    for k in ('gemm', 'conv', 'attn', 'attn_bwd', 'wgrad', 'conv_wgrad', 'gn', 'gn_bwd', 'ln', 'ln_bwd'):
        full['roofline_' + k] = dict(fam)
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline', 'loss_check', 'teacher_pass', 'step_mfma_frac'):
        assert k in line, k
    r = line['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 2500.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] is None
    assert set(line['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'} and len(line['cpu_baseline']['sample']) <= 120
    assert set(line['fracs']) == {'gemm', 'conv', 'attn', 'attn_bwd', 'wgrad', 'conv_wgrad', 'gn', 'gn_bwd', 'ln', 'ln_bwd'}
    # a single-GPU line without kernel timing is still valid
    bare = {k: v for k, v in full.items() if not k.startswith('roofline')}
    bare['comm'] = None
    assert 'roofline' not in bench.compact_line(bare)


def test_forward_mac_count_matches_the_oracle_closed_form():
    """bench.py prices `step_tflops` with sid_lsg_amd.unet.unet_forward_macs (a walk over the module tree); it equals the oracle's
    closed form (401.6 GMAC per SD1.5 forward at 64x64 latents) for every architecture and latent size."""
    from oracle.unet_ref import CONFIGS as RC, unet_forward_macs as ref
    from sid_lsg_amd.unet import CONFIGS, unet_forward_macs
    for arch in CONFIGS:
        for h in (8, 64, 96):
            assert unet_forward_macs(CONFIGS[arch], h, h) == ref(RC[arch], h, h), (arch, h)
    assert unet_forward_macs(CONFIGS['sd15'], 64, 64) == 401636720640
