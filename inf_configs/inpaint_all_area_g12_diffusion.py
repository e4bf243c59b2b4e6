"""Inference-config plugin (same two factory names and key set as the reference's
inf_configs/inpaint_all_area_g12_diffusion.py:3-73).  The TPS / inpainting keys configure the
post-forward pipeline (out of scope for the HIP hot path) and are carried as data."""
from stitch_amd.config import CfgNode as CN


def get_tps_pipline_config(cfg):
    c = CN()
    c.inpainter = "inpainter"
    c.mix_method = "inpaint_all_area"
    c.grid_h, c.grid_w = 12, 12
    c.get_pt_methods = ["advanced_uniform_multi"]
    c.tps_method = "opencv"
    c.is_plot = False
    c.limit_border_value = False
    c.inpaint_flow = False
    c.inpaint_img = True
    c.flow_pad_mode = "replicate"
    c.mesh_pad_mode = None
    c.pad_num = 4
    c.add_corner = False
    c.flow_limit = -1
    c.use_valid_on_flow = False
    c.add_meshgrid = False
    c.affine_scale = 1.0
    c.kernel_scale = 1.0
    c.use_boundary_limit = False
    c.residual_flow_use_forward = cfg.use_foward
    c.use_occ_filter = True
    c.use_border_points_mask = True
    c.do_avg_pooling = True
    c.occlusion_mask = None
    c.use_composition_when_inpaint = False
    c.output2_is_only_tps = True
    c.resize_to_area_limit_before_inpaint = 750 * 750
    return c


def get_infernce_config():
    c = CN()
    c.is_plot = False
    c.eval = "udis_eval"
    c.only_init_model = False
    c.use_composition = True
    c.composition_model_path = "./core/UDIS2/Composition/pretrained_model/epoch050_model.pth"
    c.resize_to_512 = False
    c.pad_mode = "replicate"
    c.restore_ckpt = ""
    c.test_not_use_combine_h_flow = True
    c.swap_image = False
    c.use_forward = False
    c.use_fb_consistency_mask = True
    c.use_whole_resolution = False
    return c
