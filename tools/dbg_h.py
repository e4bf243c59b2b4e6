import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
from oracle import inputs, nets, spec, geom, adapter as oad
sd = spec.seeded_state_dict(1234)
cfg,_ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
m = stitch_amd.build_model(cfg); m.load_state_dict(sd); m = m.cuda().eval()
a, b = inputs.structured_pair(512, 512, seed=7)
dev='cuda'
motion = m.predict_homo(a.cuda(), b.cuda())
print("motion", motion.cpu())
ref_motion = nets.homo_offsets(nets.W(sd,"homo_backbone."), a, b)
print("ref motion", ref_motion)
H = torch.empty((1,3,3), device=dev)
ops.dlt4(m._corners(dev, 512, 512), motion.contiguous(), H, 1, 1.0, 1.0, 8.0)
src = oad._corners(1, 512., 512.)
Href = geom.dlt4(src/8, (src+ref_motion)/8)
print("H", H.cpu(), Href)
M, Minv = m._scale_pair(dev, 64., 64.)
print(M.cpu(), Minv.cpu())
Hm = torch.empty_like(H); ops.mat3_sandwich(Minv, H, M, Hm)
print("H_mat", Hm.cpu(), oad._conj(Href, oad._scale_mat(64.,64.)))
out = ops.homo_warp(b.cuda(), Hm.view(1,9), (512,512), n_ones=3)
ref = geom.homo_transformer(torch.cat([b, torch.ones_like(b)],1), oad._conj(Href, oad._scale_mat(64.,64.)), (512,512))
d=(out.cpu()-ref).abs(); print("warp diff", d.max(), d.mean(), (d>1).float().mean())
ref2 = geom.homo_transformer(torch.cat([b, torch.ones_like(b)],1), Hm.cpu(), (512,512))
d=(out.cpu()-ref2).abs(); print("warp diff same theta", d.max(), d.mean())
