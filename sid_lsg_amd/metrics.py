"""Evaluation metrics of the distilled generator: FID and CLIP scores.

Reference: metrics/sid_metric_main.py:25-123 (registry, `calc_metric`, `report_metric`, the `fid30k_full` /
`fid_clip_30k_full` / `fid_test` / `fid_clip_test` entries), metrics/sid_fid_and_clip.py:32-74 (the Frechet distance between
the Inception feature statistics of generated images and of the real set), metrics/sid_metric_utils.py:112-188
(`FeatureStats`) and :412-510 (the generation loop: prompts through the InfiniteSampler, z ~ N(0, I) at resolution / 8,
uint8 images, 256 x 256 PIL-LANCZOS resize for the detector (reproduced bit for bit on the GPU); the CLIP score is the mean cosine of the detector's image | text halves).

What is different here (not a translation):
  * the feature statistics live ON THE GPU in fp64 (`FeatureStats.raw_mean / raw_cov` are device tensors; x^T x is one
    fp64 GEMM per batch) and ranks are merged with ONE all_reduce of (sum, sum of outer products, count) at the end instead
    of a broadcast of every feature batch from every rank;
  * trace(sqrtm(S_g S_r)) is evaluated as the sum of the square roots of the eigenvalues of the symmetric PSD matrix
    S_r^1/2 S_g S_r^1/2 (two `torch.linalg.eigh` in fp64) -- no general matrix square root, no complex arithmetic; equal to
    the reference's `np.real(trace(scipy.linalg.sqrtm(...)))` to ~1e-9 relative (tests/test_host_logic.py);
  * the generator is this package's HIP path (`sd_util.sid_sd_sampler` with `return_images=True`: UNet + VAE decoder
    kernels); the feature detectors are PLUGGABLE callables -- the reference downloads a TorchScript Inception-v3
    (`inception-2015-12-05.pt`) and pickled open_clip models, neither of which exists offline.  `load_detector(path)`
    accepts exactly those files (torch.jit / pickle) when a deployment has them; the real-set statistics come from a cached
    `.npz` (`mu`, `sigma`) or are computed from an image iterator with the same detector.
Nothing here is on the hot path.
"""
import json
import os
import pickle
import time

import numpy as np
import torch

from . import distributed as dist
from .dnnlib_util import EasyDict


# ------------------------------------------------------------------------------------------------
class FeatureStats:
    """Running mean / covariance of feature vectors (metrics/sid_metric_utils.py:112-188), accumulated in fp64 on `device`."""

    def __init__(self, capture_all=False, capture_mean_cov=False, max_items=None, device=None):
        self.capture_all, self.capture_mean_cov, self.max_items = capture_all, capture_mean_cov, max_items
        self.device = torch.device(device) if device is not None else None
        self.num_items, self.num_features = 0, None
        self.all_features, self.raw_mean, self.raw_cov = None, None, None

    def set_num_features(self, num_features, device):
        if self.num_features is not None:
            assert num_features == self.num_features
            return
        self.num_features = num_features
        self.device = self.device or device
        self.all_features = []
        self.raw_mean = torch.zeros(num_features, dtype=torch.float64, device=self.device)
        self.raw_cov = torch.zeros(num_features, num_features, dtype=torch.float64, device=self.device)

    def is_full(self):
        return self.max_items is not None and self.num_items >= self.max_items

    def append(self, x):
        x = torch.as_tensor(x)
        assert x.ndim == 2
        if self.max_items is not None and self.num_items + x.shape[0] > self.max_items:
            if self.num_items >= self.max_items:
                return
            x = x[:self.max_items - self.num_items]
        self.set_num_features(x.shape[1], x.device)
        self.num_items += x.shape[0]
        x = x.to(self.device, torch.float32)          # the reference rounds features to fp32 before accumulating in fp64
        if self.capture_all:
            self.all_features.append(x.cpu())
        if self.capture_mean_cov:
            x64 = x.to(torch.float64)
            self.raw_mean += x64.sum(0)
            self.raw_cov += x64.t() @ x64

    append_torch = append

    def merge_ranks(self, group=None):
        """Sum the accumulators over the ranks of the process group (each rank appended its own shard of the samples)."""
        if dist.get_world_size() == 1:
            return self
        assert self.capture_mean_cov and not self.capture_all
        n = torch.tensor([float(self.num_items)], dtype=torch.float64, device=self.device)
        for t in (self.raw_mean, self.raw_cov, n):
            torch.distributed.all_reduce(t, group=group)
        self.num_items = int(n.item())
        return self

    def get_all(self):
        assert self.capture_all
        return torch.cat(self.all_features, 0).numpy()

    def get_mean_cov(self):
        assert self.capture_mean_cov and self.num_items > 0
        mean = self.raw_mean / self.num_items
        cov = self.raw_cov / self.num_items - torch.outer(mean, mean)
        return mean.cpu().numpy(), cov.cpu().numpy()

    def save(self, path):
        mu, sigma = self.get_mean_cov()
        np.savez(path, mu=mu, sigma=sigma, num_items=self.num_items)


def frechet_distance(mu_gen, sigma_gen, mu_real, sigma_real):
    """|mu_g - mu_r|^2 + tr(S_g + S_r - 2 (S_g S_r)^1/2)   (metrics/sid_fid_and_clip.py:65-67).
    tr((S_g S_r)^1/2) = sum_i sqrt(lambda_i(S_r^1/2 S_g S_r^1/2)): both factors are symmetric PSD, so two fp64 `eigh`
    give it without a general matrix square root."""
    mu_g, mu_r = (torch.as_tensor(np.asarray(m), dtype=torch.float64) for m in (mu_gen, mu_real))
    s_g, s_r = (torch.as_tensor(np.asarray(s), dtype=torch.float64) for s in (sigma_gen, sigma_real))
    w, v = torch.linalg.eigh((s_r + s_r.t()) * 0.5)
    root_r = (v * w.clamp_min(0).sqrt()) @ v.t()
    mid = root_r @ ((s_g + s_g.t()) * 0.5) @ root_r
    tr_sqrt = torch.linalg.eigvalsh((mid + mid.t()) * 0.5).clamp_min(0).sqrt().sum()
    return float((mu_g - mu_r).square().sum() + torch.trace(s_g) + torch.trace(s_r) - 2.0 * tr_sqrt)


def clip_score_from_features(features):
    """features [N, 2F] = image | text halves (already normalised by the detector): mean cosine (sid_metric_utils.py:503-504)."""
    f = torch.as_tensor(features)
    img, txt = f.tensor_split((f.shape[1] // 2,), 1)
    return float((img * txt).sum(-1).mean())


# ------------------------------------------------------------------------------------------------
def load_detector(path, device):
    """A feature detector file of the reference's kinds: TorchScript (`inception-2015-12-05.pt`, called as
    `detector(uint8 NCHW images, return_features=True)`) or a pickled module (open_clip / CLIP wrappers, called with
    `texts=..., div255=True`).  Offline there are none: callers may pass any callable instead."""
    if callable(path):
        return path
    if not path or not os.path.isfile(path):
        raise FileNotFoundError(f'feature detector {path!r} not found: FID / CLIP metrics need the Inception / CLIP files the reference '
                                'downloads (metrics/sid_fid_and_clip.py:36, sid_metric_utils.py:456); pass a local file or a callable')
    try:
        return torch.jit.load(path, map_location=device).eval()
    except Exception:
        with open(path, 'rb') as f:
            return pickle.load(f).to(device).eval()


_LANCZOS_CACHE = {}


def _lanczos_coefficients(in_size, out_size):
    """The integer filter bank of Pillow's 8-bit LANCZOS resampling, [out_size, in_size] int64 (Pillow src/libImaging/Resample.c:
    `precompute_coeffs` with the a = 3 windowed sinc -- support 3 * max(scale, 1) input pixels around (x + 0.5) * scale, weights
    normalised to sum 1 -- then `normalize_coeffs_8bpc`: round-half-away to 22 fractional bits)."""
    import math
    key = (in_size, out_size)
    if key in _LANCZOS_CACHE:
        return _LANCZOS_CACHE[key]
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    K = np.zeros((out_size, in_size), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        arg = (np.arange(xmin, xmax) - center + 0.5) / fscale
        with np.errstate(invalid='ignore', divide='ignore'):
            sinc = lambda t: np.where(t == 0, 1.0, np.sin(np.pi * t) / (np.pi * t))      # noqa: E731
            w = np.where((arg >= -3.0) & (arg < 3.0), sinc(arg) * sinc(arg / 3.0), 0.0)
        tot = w.sum()
        if tot != 0:
            w = w / tot
        K[xx, xmin:xmax] = np.where(w < 0, np.trunc(-0.5 + w * (1 << 22)), np.trunc(0.5 + w * (1 << 22))).astype(np.int64)
    _LANCZOS_CACHE[key] = K
    return K


def resize_for_detector(images_u8, size=256):
    """uint8 NCHW -> uint8 NCHW at size x size with the arithmetic of the reference's `resize_images_in_tensor`
    (sid_metric_utils.py:353-375: per image `PIL.Image.resize((256, 256), Image.LANCZOS)`, back to uint8) -- on the device, for the
    whole batch: Pillow's two passes (horizontal, then vertical, each rounded to 8 bits) with its 22-bit fixed-point coefficients.
    Sums of 8-bit pixels times 22-bit integers are exact in fp64, so two fp64 GEMMs reproduce the integer pipeline bit for bit
    (pinned against PIL itself: tests/golden/metrics_ref.npz, tests/test_host_logic.py)."""
    if images_u8.dtype != torch.uint8 or images_u8.ndim != 4:
        raise ValueError('resize_for_detector: uint8 NCHW images')
    H, W = images_u8.shape[-2:]
    if H == size and W == size:
        return images_u8
    dev = images_u8.device
    kh = torch.from_numpy(_lanczos_coefficients(W, size)).to(dev, torch.float64)
    kv = torch.from_numpy(_lanczos_coefficients(H, size)).to(dev, torch.float64)
    half, one = float(1 << 21), float(1 << 22)
    t = torch.matmul(images_u8.to(torch.float64), kh.t())                      # [N, C, H, size]
    t = torch.floor((t + half) / one).clamp_(0, 255)
    t = torch.matmul(kv, t)                                                    # [N, C, size, size]
    t = torch.floor((t + half) / one).clamp_(0, 255)
    return t.to(torch.uint8)


class _PromptList:
    """A list of prompts with the dataset item contract `(image, text)` (for callers that hold plain strings)."""

    def __init__(self, prompts):
        self.prompt_list = list(prompts)

    def __len__(self):
        return len(self.prompt_list)

    def __getitem__(self, i):
        return None, self.prompt_list[i]


class MetricOptions:
    """What a metric needs (reference: sid_metric_utils.MetricOptions): a generator `G(latents=, contexts=, init_timesteps=)`
    returning images in [-1, 1], the prompt source, the detectors, the real-set statistics.
    Prompt source: `dataset_kwargs` (what the reference passes, sid_training_loop.py:636: the evaluation caption set of
    `--data`, built by class name) or `dataset` (an object with that item contract) or `prompts` (a list of strings).  The
    evaluation order is the reference's: `InfiniteSampler(dataset, rank, num_gpus, seed=0)` (sid_metric_utils.py:420)."""

    def __init__(self, G, prompts=None, resolution=512, init_timestep=625, detector=None, real_stats=None, open_clip_detector=None,
                 clip_score_fn=None, device=None, seed=0, batch_gen=4, detector_size=256, progress=None, dataset_kwargs=None,
                 dataset=None):
        from .dnnlib_util import construct_class_by_name
        if dataset is None and dataset_kwargs:
            dataset = construct_class_by_name(**dataset_kwargs)
        if dataset is None:
            if prompts is None:
                raise ValueError('metrics need a prompt source: dataset_kwargs, dataset or prompts')
            dataset = _PromptList(prompts)
        self.dataset = dataset
        self.G, self.resolution, self.init_timestep = G, resolution, init_timestep
        self.detector, self.real_stats, self.open_clip_detector, self.clip_score_fn = detector, real_stats, open_clip_detector, clip_score_fn
        self.device = torch.device(device if device is not None else 'cuda')
        self.seed, self.batch_gen, self.detector_size, self.progress = seed, batch_gen, detector_size, progress
        self.rank, self.num_gpus = dist.get_rank(), dist.get_world_size()


def generator_feature_stats(opts, num_gen, compute_clip=False):
    """sid_metric_utils.py:412-510: this rank's share of `num_gen` samples -- prompts in the order of the reference's
    `InfiniteSampler(dataset, rank, num_gpus, seed=0)` (shuffled, rank-strided; :420), z ~ N(0, I) from a per-rank generator --
    through G, the detector and (optionally) the CLIP detectors."""
    from .data import InfiniteSampler
    order = iter(InfiniteSampler(opts.dataset, rank=opts.rank, num_replicas=opts.num_gpus, seed=0))
    detector = load_detector(opts.detector, opts.device)
    oc = load_detector(opts.open_clip_detector, opts.device) if (compute_clip and opts.open_clip_detector is not None) else None
    stats = FeatureStats(capture_mean_cov=True, max_items=None, device=opts.device)
    gen = torch.Generator(device=opts.device).manual_seed(opts.seed * opts.num_gpus + opts.rank)
    lat = opts.resolution // 8
    mine = list(range(opts.rank, num_gen, opts.num_gpus))          # global sample indices of this rank
    oc_scores, clip_scores = [], []
    for i in range(0, len(mine), opts.batch_gen):
        idx = mine[i:i + opts.batch_gen]
        texts = [opts.dataset[next(order)][1] for _ in idx]
        z = torch.randn([len(idx), 4, lat, lat], device=opts.device, generator=gen)
        with torch.no_grad():
            img = opts.G(latents=z, contexts=texts, init_timesteps=opts.init_timestep * torch.ones(len(idx), device=opts.device, dtype=torch.long))
        img = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8)
        if img.shape[1] == 1:
            img = img.repeat(1, 3, 1, 1)
        img = resize_for_detector(img, opts.detector_size)
        with torch.no_grad():
            stats.append(detector(img, return_features=True))
            if compute_clip:
                if opts.clip_score_fn is not None:
                    clip_scores.append(torch.as_tensor(opts.clip_score_fn(img, texts)).float().flatten().cpu())
                if oc is not None:
                    oc_scores.append(torch.tensor([clip_score_from_features(oc(img, texts=texts, div255=True))] * len(idx)))
        if opts.progress is not None:
            opts.progress(stats.num_items * opts.num_gpus, num_gen)
    stats.merge_ranks()

    def mean_over_ranks(parts):
        if not parts:
            return float('nan')
        v = torch.cat(parts).double()
        t = torch.tensor([float(v.sum()), float(v.numel())], dtype=torch.float64, device=opts.device)
        if opts.num_gpus > 1:
            torch.distributed.all_reduce(t)
        return float(t[0] / t[1])
    return stats, mean_over_ranks(oc_scores), mean_over_ranks(clip_scores)


def load_real_stats(real_stats):
    """(mu, sigma) of the real set: a cached .npz / .pkl of the reference's FeatureStats dict, or a (mu, sigma) pair."""
    if isinstance(real_stats, (tuple, list)):
        return np.asarray(real_stats[0]), np.asarray(real_stats[1])
    if isinstance(real_stats, str) and real_stats.endswith('.npz'):
        d = np.load(real_stats)
        return d['mu'], d['sigma']
    if isinstance(real_stats, str):
        with open(real_stats, 'rb') as f:
            s = pickle.load(f)
        mean = s['raw_mean'] / s['num_items']
        return mean, s['raw_cov'] / s['num_items'] - np.outer(mean, mean)
    raise ValueError('real_stats: a (mu, sigma) pair, a .npz with mu / sigma, or a pickled FeatureStats of the reference')


def compute_fid_and_clip(opts, num_gen, compute_clip=False):
    """metrics/sid_fid_and_clip.py:32-74."""
    mu_real, sigma_real = load_real_stats(opts.real_stats)
    stats, open_clip_score, clip_score = generator_feature_stats(opts, num_gen, compute_clip)
    mu_gen, sigma_gen = stats.get_mean_cov()
    fid = frechet_distance(mu_gen, sigma_gen, mu_real, sigma_real)
    return (fid, open_clip_score, clip_score) if compute_clip else fid


# ------------------------------------------------------------------------------------------------
_metric_dict = {}


def register_metric(fn):
    _metric_dict[fn.__name__] = fn
    return fn


def is_valid_metric(metric):
    return metric in _metric_dict


def list_valid_metrics():
    return list(_metric_dict.keys())


@register_metric
def fid30k_full(opts):
    return dict(fid30k_full=compute_fid_and_clip(opts, 30000), open_clipscore_30k=float('nan'), clipscore30k=float('nan'))


@register_metric
def fid_clip_30k_full(opts):
    fid, oc, cs = compute_fid_and_clip(opts, 30000, compute_clip=True)
    return dict(fid30k_full=fid, open_clipscore_30k=oc, clipscore30k=cs)


@register_metric
def fid_test(opts):
    return dict(fid30k_full=compute_fid_and_clip(opts, max(1, getattr(opts, 'num_test', 1))), open_clipscore_30k=float('nan'), clipscore30k=float('nan'))


@register_metric
def fid_clip_test(opts):
    fid, oc, cs = compute_fid_and_clip(opts, max(1, getattr(opts, 'num_test', 1)), compute_clip=True)
    return dict(fid30k_full=fid, open_clipscore_30k=oc, clipscore30k=cs)


def calc_metric(metric, **kwargs):
    """sid_metric_main.py:46-72: run one registered metric, return the decorated result dict."""
    if not is_valid_metric(metric):
        raise ValueError(f'unknown metric {metric!r}; valid: {list_valid_metrics()}')
    num_test = kwargs.pop('num_test', None)
    opts = MetricOptions(**kwargs)
    if num_test is not None:
        opts.num_test = num_test
    t0 = time.time()
    results = _metric_dict[metric](opts)
    total = time.time() - t0
    return EasyDict(results=EasyDict(results), metric=metric, total_time=total, total_time_str=f'{total:.1f}s', num_gpus=opts.num_gpus)


def report_metric(result_dict, run_dir=None, snapshot_pkl=None, alpha=None, num_steps_eval=None):
    """sid_metric_main.py:82-99: one JSON line on stdout and in `metric-<name>[-alpha-..][-num_steps_eval-..].jsonl`."""
    metric = result_dict['metric']
    if run_dir is not None and snapshot_pkl is not None:
        snapshot_pkl = os.path.relpath(snapshot_pkl, run_dir)
    line = json.dumps(dict(result_dict, snapshot_pkl=snapshot_pkl, timestamp=time.time()))
    dist.print0(line)
    if run_dir is not None and os.path.isdir(run_dir) and dist.get_rank() == 0:
        name = f'metric-{metric}'
        if alpha is not None:
            name += f'-alpha-{alpha:03f}'
            if num_steps_eval is not None and num_steps_eval != 1:
                name += f'-num_steps_eval-{num_steps_eval:02d}'
        with open(os.path.join(run_dir, name + '.jsonl'), 'at') as f:
            f.write(line + '\n')
    return line
