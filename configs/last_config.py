"""Model hyper-parameters of the stitching path (the live subset of the reference's
configs/last_config.py:1 -- training / logging keys carry no meaning for inference and are omitted)."""
config_dict = {
    "name": "udis",
    "transformer": "percostformer3",
    "homo_backbone": "udis2",
    "flow_backbone": "flowformerpp",
    "image_size": [512, 512],
    "batch_size": 1,
    "only_homo": False,
    "detach_H": False,
    "detach_flow": False,
    "use_foward": False,            # (sic) the key the reference ships; FlowHomoAdpater reads `use_forward`
    "use_combine_h_flow": False,
    "use_fb_consistency_mask": True,
    "percostformer3": {
        "pe": "linear", "gma": "GMA", "cnet": "twins", "fnet": "twins", "gt_r": 15, "r_16": -1, "no_sc": False,
        "fix_pe": False, "dropout": 0, "use_rpe": False, "droppath": 0, "pic_size": [368, 496, 368, 496],
        "pretrain": False, "use_patch": False, "cross_attn": "all", "del_layers": True, "flow_or_pe": "and",
        "patch_size": 8, "vert_c_dim": 64, "patch_embed": "single", "detach_local": False, "decoder_depth": 12,
        "encoder_depth": 3, "pretrain_mode": False, "quater_refine": False, "use_convertor": False,
        "cost_heads_num": 1, "cost_latent_dim": 128, "cost_encoder_res": True, "query_latent_dim": 64,
        "encoder_latent_dim": 256, "cost_latent_input_dim": 64, "cost_latent_token_num": 8,
        "vertical_encoder_attn": "twins",
    },
}
