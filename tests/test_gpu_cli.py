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
