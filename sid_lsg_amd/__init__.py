"""sid_lsg_amd: MI355X-native (gfx950) implementation of the SiD-LSG distillation inner step.

Scope: the score-identity inner step of mingyuanzhou/SiD-LSG
(training/sid_training_loop.py:383-571 + training/sid_sd_util.py) as hand-written HIP kernels
behind the reference's own seams: `load_sd15` / `sid_sd_sampler` / `sid_sd_denoise` /
`training_loop`, `custom_ops.get_plugin`, `dnnlib.util.construct_class_by_name`, `bias_act`.
(The directory is named sid_lsg_amd -- with an underscore -- so that it is importable.)
"""
__version__ = '0.1.0'
