#!/usr/bin/env python
"""One-step text-to-image generation with a distilled generator (counterpart of the reference's `generate_onestep.py`).

    torchrun --standalone --nproc_per_node=8 generate_onestep.py --network runs/00000-.../network-snapshot-1.000000-000500.pkl \\
        --outdir out --seeds 0-63 --batch 16 --text_prompts prompts.txt --repo_id /models/stable-diffusion-v1-5

Same options and file layout as the reference (`generate_onestep.py:113-125, 218-311`): image i uses prompt line i and the
latent drawn from torch.Generator(seed_i) (`StackedRandomGenerator`), x_hat = G(z; t_init) in one UNet evaluation,
`vae.decode(x_hat / scaling_factor)`, uint8 `(img*127.5+128).clip(0,255)` PNG named `<seed:06d>.png`.  Batches are strided
over ranks.  `--repo_id` must be a local diffusers-layout directory (text encoder, tokenizer, VAE) or `random:<arch>`.
"""
import os
import pickle
import re

import click
import numpy as np
import torch

from sid_lsg_amd import distributed as dist
from sid_lsg_amd.sd_util import load_sd15, sid_sd_sampler


class StackedRandomGenerator:
    """One torch.Generator per sample, so that image i does not depend on the batch it is generated in."""

    def __init__(self, device, seeds):
        self.generators = [torch.Generator(device).manual_seed(int(s) % (1 << 32)) for s in seeds]

    def randn(self, size, **kw):
        assert size[0] == len(self.generators)
        return torch.stack([torch.randn(size[1:], generator=g, **kw) for g in self.generators])


def parse_int_list(s):
    """'1,2,5-10' -> [1, 2, 5, 6, 7, 8, 9, 10]"""
    if isinstance(s, list):
        return s
    out = []
    for part in s.split(','):
        m = re.fullmatch(r'(\d+)-(\d+)', part)
        out.extend(range(int(m.group(1)), int(m.group(2)) + 1) if m else [int(part)])
    return out


def read_prompts(path):
    with open(path, 'rt') as f:
        return [line.strip() for line in f if line.strip()]


def save_png(path, hwc_uint8):
    try:
        import PIL.Image
        PIL.Image.fromarray(hwc_uint8, 'RGB').save(path)
    except ImportError:                    # no Pillow: minimal PNG writer (zlib + CRC), enough for RGB8
        import struct
        import zlib
        h, w, _ = hwc_uint8.shape
        raw = b''.join(b'\x00' + hwc_uint8[y].tobytes() for y in range(h))

        def chunk(tag, data):
            return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
        with open(path, 'wb') as f:
            f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
                    chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


@click.command()
@click.option('--network', 'network_pkl', type=str, required=True, metavar='PATH', help='Network snapshot pickle')
@click.option('--outdir', type=str, required=True, metavar='DIR', help='Where to save the output images')
@click.option('--seeds', type=parse_int_list, default='0-63', show_default=True, metavar='LIST', help='Random seeds (e.g. 1,2,5-10)')
@click.option('--subdirs', is_flag=True, help='Create subdirectory for every 1000 seeds')
@click.option('--batch', 'max_batch_size', type=click.IntRange(min=1), default=16, show_default=True, help='Maximum batch size')
@click.option('--num', 'num_fid_samples', type=click.IntRange(min=1), default=30000, show_default=True, help='Maximum number of images')
@click.option('--init_timestep', type=click.IntRange(min=0), default=625, show_default=True, help='t_init, in [0,999]')
@click.option('--text_prompts', type=str, default='prompts/captions.txt', show_default=True, help='Prompt file, one per line')
@click.option('--repo_id', type=str, default='runwayml/stable-diffusion-v1-5', show_default=True, help='Local diffusers directory or random:<arch>')
@click.option('--use_fp16', type=bool, default=True, show_default=True, help='Accepted for compatibility (compute is bf16)')
@click.option('--enable_compress_npz', type=bool, default=False, show_default=True, help='Also write the batch as images.npz')
@click.option('--num_steps_eval', type=click.IntRange(min=0), default=1, show_default=True, help='Generation steps (1 = one-step)')
@click.option('--custom_seed', type=bool, default=False, show_default=True, help='Prompt i <-> i-th seed of the list instead of seed value')
def main(network_pkl, outdir, seeds, subdirs, max_batch_size, num_fid_samples, init_timestep, text_prompts, repo_id, use_fp16,
         enable_compress_npz, num_steps_eval, custom_seed):
    dist.init()
    device = torch.device('cuda')
    rank, world = dist.get_rank(), dist.get_world_size()
    captions = read_prompts(text_prompts)
    seeds = seeds[:num_fid_samples]
    num_batches = ((len(seeds) - 1) // (max_batch_size * world) + 1) * world
    index = torch.arange(len(seeds)) if custom_seed else torch.as_tensor(seeds)
    rank_batches = index.tensor_split(num_batches)[rank::world]

    if world > 1 and rank != 0:
        torch.distributed.barrier()                    # rank 0 touches the files first
    dist.print0(f'Loading network from "{network_pkl}"...')
    with open(network_pkl, 'rb') as f:
        G_ema = pickle.load(f)['ema'].to(device)
    G_ema.eval().requires_grad_(False)
    _, vae, sched, text_encoder, tokenizer = load_sd15(repo_id, repo_id, device, torch.bfloat16)
    del _
    if world > 1 and rank == 0:
        torch.distributed.barrier()
    if num_steps_eval > 1:
        outdir = f'{outdir}_numstep{num_steps_eval}'

    lat = 64
    dist.print0(f'Generating {len(seeds)} images to "{outdir}"...')
    for batch in rank_batches:
        if world > 1:
            torch.distributed.barrier()
        if len(batch) == 0:
            continue
        batch = [int(b) for b in batch]
        batch_seeds = [seeds[i] for i in batch] if custom_seed else batch
        z = StackedRandomGenerator(device, batch_seeds).randn([len(batch), 4, lat, lat], device=device)
        prompts = [captions[i % len(captions)] for i in batch]
        with torch.no_grad():
            images = sid_sd_sampler(unet=G_ema, latents=z, contexts=prompts,
                                    init_timesteps=init_timestep * torch.ones(len(batch), device=device, dtype=torch.long),
                                    noise_scheduler=sched, text_encoder=text_encoder, tokenizer=tokenizer, resolution=512,
                                    dtype=torch.bfloat16, return_images=True, vae=vae, num_steps=1, train_sampler=False,
                                    num_steps_eval=num_steps_eval)
        arr = (images.float() * 127.5 + 128).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        for key, img in zip(batch, arr):
            d = os.path.join(outdir, f'{key - key % 1000:06d}') if subdirs else outdir
            os.makedirs(d, exist_ok=True)
            save_png(os.path.join(d, f'{key:06d}.png'), np.ascontiguousarray(img))
        if enable_compress_npz:
            np.savez_compressed(os.path.join(outdir, f'images_{batch[0]:06d}.npz'), images=arr)
    if world > 1:
        torch.distributed.barrier()
    dist.print0('Done.')


if __name__ == '__main__':
    main()
