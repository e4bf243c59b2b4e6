"""Stand-in for the reference's neural inpainters (core/inference/mix_methods/utils/{transref_,}inpainter.py: fetched
weights and third-party CUDA ops, out of scope): same protocol, returns the control image (or the input) unchanged, so the
post-pipeline runs end to end and the holes keep what `mix_fn` filled from image 1."""


class Inpainter:
    def __init__(self):
        self.name = "passthrough_inpainter"

    def inpaint(self, init_image_tensor, mask_image_tensor, control_image_tensor=None, prompt="", resize_to_area_limit_before_inpaint=False):
        src = control_image_tensor if control_image_tensor is not None else init_image_tensor
        return src.clone()


inpainter = Inpainter()
