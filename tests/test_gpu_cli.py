"""End to end through the two command lines: sid_train.py (distillation, snapshot) -> generate_onestep.py (PNG files)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _png_pixels(path):
    import PIL.Image
    return np.asarray(PIL.Image.open(path).convert('RGB'))


def test_train_then_generate(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from click.testing import CliRunner
    import generate_onestep
    import sid_train
    (tmp_path / 'aesthetics_6_plus.txt').write_text('\n'.join(f'prompt number {i}' for i in range(40)) + '\n')
    runs = tmp_path / 'runs'
    res = CliRunner().invoke(sid_train.main, [
        '--outdir', str(runs), '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny', '--seed', '1', '--batch', '4',
        '--batch-gpu', '2', '--duration', '0.00004', '--ema', '0.00001', '--tick', '1', '--snap', '1', '--dump', '1',
        '--cfg_train_fake', '1.5', '--cfg_eval_fake', '1.5', '--cfg_eval_real', '1.5', '--resolution', '128'],
        catch_exceptions=False)
    assert res.exit_code == 0, res.output
    run_dir = glob.glob(str(runs / '00000-*'))[0]
    snaps = sorted(glob.glob(os.path.join(run_dir, 'network-snapshot-*.pkl')))
    assert snaps and glob.glob(os.path.join(run_dir, 'training-state-*.pt'))
    assert os.path.isfile(os.path.join(run_dir, 'training_options.json'))

    # --transfer: a second run starts its generator from the snapshot; --resume: from the training state
    states = sorted(glob.glob(os.path.join(run_dir, 'training-state-*.pt')))
    for extra in (['--transfer', snaps[-1]], ['--resume', states[-1]]):
        res = CliRunner().invoke(sid_train.main, [
            '--outdir', str(runs), '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny', '--seed', '2', '--batch', '4',
            '--batch-gpu', '2', '--duration', '0.00003', '--ema', '0.00001', '--tick', '1', '--snap', '50', '--dump', '50',
            '--resolution', '128'] + extra, catch_exceptions=False)
        assert res.exit_code == 0, res.output
    assert len(glob.glob(str(runs / '0000[0-9]-*'))) == 3

    outs = []
    for batch in (2, 4):
        out = tmp_path / f'img_b{batch}'
        res = CliRunner().invoke(generate_onestep.main, [
            '--network', snaps[-1], '--outdir', str(out), '--seeds', '0-3', '--batch', str(batch),
            '--text_prompts', str(tmp_path / 'aesthetics_6_plus.txt'), '--repo_id', 'random:tiny'], catch_exceptions=False)
        assert res.exit_code == 0, res.output
        files = sorted(glob.glob(str(out / '*.png')))
        assert [os.path.basename(f) for f in files] == [f'{i:06d}.png' for i in range(4)]
        outs.append([_png_pixels(f) for f in files])
    # per-sample seeding: an image does not depend on the batch it was generated in (up to a few uint8 levels: the batch
    # size changes tile / split-K choices, i.e. the fp32 summation order)
    for a, b in zip(outs[0], outs[1]):
        assert a.shape == b.shape == (512, 512, 3)
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 4
    with open(glob.glob(str(tmp_path / 'img_b2' / '*.png'))[0], 'rb') as f:
        assert f.read(8) == b'\x89PNG\r\n\x1a\n'


def test_train_with_fid_metric(tmp_path):
    """`sid_train.py --metrics fid_test` -- at the snapshot ticks the EMA generator runs through the
    one-step sampler + VAE decoder (HIP kernels), a TorchScript feature detector with the reference's calling convention
    (`detector(uint8 NCHW, return_features=True)`, here a tiny scripted conv net standing in for inception-2015-12-05.pt) and
    the Frechet distance against cached real-set statistics; the result lands in metric-fid_test-alpha-*.jsonl and the stats."""
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    import json

    import numpy as np
    from click.testing import CliRunner
    import sid_train

    class Detector(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.conv = torch.nn.Conv2d(3, 16, 8, stride=8)

        def forward(self, img: torch.Tensor, return_features: bool = True) -> torch.Tensor:
            return self.conv(img.to(torch.float32) / 255.0).mean(dim=(2, 3))
    det_path, stat_path = str(tmp_path / 'detector.pt'), str(tmp_path / 'real_stats.npz')
    torch.jit.script(Detector()).save(det_path)
    np.savez(stat_path, mu=np.zeros(16), sigma=np.eye(16) * 0.01)
    (tmp_path / 'aesthetics_6_plus.txt').write_text('\n'.join(f'prompt number {i}' for i in range(40)) + '\n')
    runs = tmp_path / 'runs'
    res = CliRunner().invoke(sid_train.main, [
        '--outdir', str(runs), '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny', '--seed', '1', '--batch', '4',
        '--batch-gpu', '2', '--duration', '0.00002', '--ema', '0.00001', '--tick', '1', '--snap', '1', '--dump', '50',
        '--resolution', '128', '--metrics', 'fid_test', '--metric_pt_path', det_path, '--data_stat', stat_path], catch_exceptions=False)
    assert res.exit_code == 0, res.output
    run_dir = glob.glob(str(runs / '00000-*'))[0]
    files = glob.glob(os.path.join(run_dir, 'metric-fid_test-alpha-*.jsonl'))
    assert files, os.listdir(run_dir)
    rows = [json.loads(ln) for ln in open(files[0])]
    assert rows and all(np.isfinite(r['results']['fid30k_full']) and r['results']['fid30k_full'] > 0 for r in rows)
    assert all(r['snapshot_pkl'].startswith('network-snapshot-') for r in rows)
    stats = [json.loads(ln) for ln in open(glob.glob(os.path.join(run_dir, 'stats_*.jsonl'))[0])]
    assert any('Metrics/fid30k_full' in r for r in stats)
    # --train_mode 0 (sid_training_loop.py:680-745): evaluate the snapshot the run wrote, 1 / 2 / 4 generation steps, on the caption set
    # of --data, results as `<metric><kimg>_<steps>.txt` next to the run directory in the reference's `key: value` format
    snaps = sorted(glob.glob(os.path.join(run_dir, 'network-snapshot-*.pkl')))
    assert snaps
    caps = tmp_path / 'coco_captions.txt'
    caps.write_text('\n'.join(f'evaluation caption {i}' for i in range(9)) + '\n')
    ev = CliRunner().invoke(sid_train.main, [
        '--outdir', str(runs), '--data_prompt_text', str(tmp_path), '--data', str(caps), '--sd_model', 'random:tiny', '--seed', '1', '--resolution', '128',
        '--train_mode', '0', '--network_pkl', snaps[-1], '--metrics', 'fid_test', '--metric_pt_path', det_path, '--data_stat', stat_path],
        catch_exceptions=False)
    assert ev.exit_code == 0, ev.output
    kimg = snaps[-1][-10:-4]
    for steps in (1, 2, 4):
        txt = os.path.join(str(runs), f'fid_test{kimg}_{steps}.txt')
        assert os.path.isfile(txt), os.listdir(str(runs))
        body = open(txt).read()
        assert body.startswith('results: ') and 'fid30k_full' in body and 'metric: fid_test' in body
    none = CliRunner().invoke(sid_train.main, ['--outdir', str(runs), '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny',
                                               '--train_mode', '0', '--metrics', 'fid_test', '--metric_pt_path', det_path, '--data_stat', stat_path])
    assert none.exit_code != 0 and '--network_pkl' in none.output
    # missing files are refused up front
    bad = CliRunner().invoke(sid_train.main, ['--outdir', str(runs), '--data_prompt_text', str(tmp_path), '--sd_model', 'random:tiny',
                                              '--metrics', 'fid30k_full', '--metric_pt_path', '/nonexistent.pt', '-n'])
    assert bad.exit_code != 0 and '--metric_pt_path' in bad.output


@pytest.mark.gpu
def test_metrics_pipeline_matches_reference_golden_on_gpu(golden_dir):
    """SURVEY section 8(f4) on the device: prompt order, uint8 conversion, the PIL-LANCZOS resize (fp64 GEMMs on the GPU), the fp64
    feature statistics on the GPU and the eigenvalue form of the Frechet distance against vectors from the REFERENCE's own
    evaluation code (tests/golden/metrics_ref.npz, oracle/make_goldens.py::gen_metrics)."""
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from test_host_logic import replay_metrics_golden
    replay_metrics_golden(golden_dir, 'cuda')
