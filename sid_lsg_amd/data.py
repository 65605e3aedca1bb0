"""Prompt stream of the distillation loop.

  * `PromptDataset`: item contract of the reference's training/aesthetics_dataset.py::ImageDataset
    (`(dummy_image, prompt)`; file lookup order aesthetics_6_plus.txt, aesthetics_625_plus.txt,
    aesthetics_65_plus.txt under `path`; attributes name / resolution), without the blobfile/PIL imports.
  * `CaptionDataset`: the caption side of the reference's training/mscoco_dataset.py::ImageDataset (the COCO-2014 validation
    set of `--data`, read only by the metrics: metrics/sid_metric_utils.py:419-421 draws the evaluation prompts from it through
    `InfiniteSampler(seed=0)`): every image file that has a same-named `.txt` next to it, in the reference's sorted recursive
    order; the pixels are never decoded here (the generator metrics only use the text).
  * `InfiniteSampler`: rank-strided, windowed-shuffle index stream with the semantics of
    torch_utils/misc.py:110-141 (it defines which prompt each rank sees at each step); pinned against the
    reference by tests/golden/sampler.npz.
"""
import os

import numpy as np
import torch


class PromptDataset(torch.utils.data.Dataset):
    FILES = ('aesthetics_6_plus.txt', 'aesthetics_625_plus.txt', 'aesthetics_65_plus.txt')

    def __init__(self, path, resolution=512, random_crop=False, random_flip=0.0, prompt_only=True):
        assert prompt_only, 'prompt_only must be True for the prompt dataset'
        self.name, self.resolution = 'aesthetics', resolution
        full = path
        if os.path.isdir(path):
            for fn in self.FILES:
                full = os.path.join(path, fn)
                if os.path.exists(full):
                    break
        with open(full, 'rt') as f:
            self.prompt_list = [row.strip('\n') for row in f]
        if not self.prompt_list:
            raise IOError(f'no prompts in {full}')

    def __len__(self):
        return len(self.prompt_list)

    def __getitem__(self, idx):
        return torch.zeros(1, 4, 4), self.prompt_list[idx]


class CaptionDataset(torch.utils.data.Dataset):
    """(dummy_image, caption) items of an image + caption directory (training/mscoco_dataset.py:11-44), or of a plain text
    file with one caption per line."""
    IMAGE_EXT = ('jpg', 'jpeg', 'png', 'gif', 'webp')

    def __init__(self, path, resolution=512, random_crop=False, random_flip=0.0):
        self.name, self.resolution = 'MSCOCO-2014', resolution
        if os.path.isfile(path):
            with open(path, 'rt') as f:
                self.captions = [row.strip() for row in f if row.strip()]
        else:
            self.captions = []
            for txt in self._caption_files(path):
                with open(txt, 'rt') as f:
                    self.captions.append(f.read().strip())
        if not self.captions:
            raise IOError(f'no captions under {path}')

    @classmethod
    def _caption_files(cls, path):
        out = []
        for entry in sorted(os.listdir(path)):
            full = os.path.join(path, entry)
            parts = entry.split('.')
            if parts[-1].strip().lower() in cls.IMAGE_EXT and len(parts) > 1:
                txt = os.path.join(path, parts[0] + '.txt')
                if os.path.exists(txt):
                    out.append(txt)
            elif os.path.isdir(full):
                out.extend(cls._caption_files(full))
        return out

    def __len__(self):
        return len(self.captions)

    def __getitem__(self, idx):
        return torch.zeros(1, 4, 4), self.captions[idx]


class InfiniteSampler:
    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        self.n, self.rank, self.num_replicas = len(dataset), rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(self.n)
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield int(order[i])
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1


def prompt_batches(dataset, sampler, batch):
    it = iter(sampler)
    while True:
        yield [dataset[next(it)][1] for _ in range(batch)]
