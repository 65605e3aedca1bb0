"""CPU oracle for the SiD-LSG distillation inner step.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 (CPU) restatement of the algorithm on the
hot path named by BASELINE.json.  It is the *checker*:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `sid_lsg_amd/` imports it, and the
product path raises if its HIP library is missing instead of falling back here.

Pinning status (see DESIGN.md section "Oracle"):
  * glue / loop math (sampler, CFG denoise, both losses, Adam(beta1=0), EMA,
    NaN rules, accumulation): PINNED against the reference itself, imported in
    the authoring container by `oracle/ref_harness.py`; golden vectors are
    committed under `tests/golden/` together with `oracle/make_goldens.py`.
  * UNet / DDPM-scheduler arithmetic: the reference delegates these to
    diffusers==0.27.2 (sid_lsg_environment.yml:12), which is NOT in
    /root/reference and not installed offline -> **parity unpinned** for the
    UNet numerics; structurally pinned by exact parameter totals
    (859 520 964 / 865 910 724) and the diffusers state_dict key set.
  * CLIP text encoder: pinned against transformers.CLIPTextModel (installed
    here) by weight sharing in tests/test_text_encoder.py.
"""
