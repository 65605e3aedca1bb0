"""Prompt stream of the distillation loop.

  * `PromptDataset`: item contract of the reference's training/aesthetics_dataset.py::ImageDataset
    (`(dummy_image, prompt)`; file lookup order aesthetics_6_plus.txt, aesthetics_625_plus.txt,
    aesthetics_65_plus.txt under `path`; attributes name / resolution), without the blobfile/PIL imports.
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
