"""`mix_fn` plug-ins of the TPS post-pipeline (reference: core/inference/mix_methods/<name>.py, selected by
`TPS_PIPELINE_CONFIG.mix_method`, out.py:235): same module names, same `mix_fn` signature and return tuple.  The mask /
image algebra runs in HIP kernels; the inpainter is whatever object the caller passes (`.name`, `.inpaint(...)`, the
reference's protocol: core/inference/mix_methods/utils/transref_inpainter.py:16,37).  The neural inpainters themselves
(TransRef, diffusion, GAN) are out of scope; `utils.passthrough_inpainter` stands in for them."""
