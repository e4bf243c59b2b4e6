import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
from oracle import inputs, nets, spec, adapter as oad
sd = spec.seeded_state_dict(1234)
cfg,_ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
m = stitch_amd.build_model(cfg); m.load_state_dict(sd); m = m.cuda().eval()
g = np.load('/root/repo/tests/golden/e2e_eval_512.npz')
a, b = inputs.structured_pair(512, 512, seed=7)
o = m(a.cuda(), b.cuda(), type="test_eval")
print("output_H sub err p99/max", np.percentile(np.abs(o["output_H"][..., ::8, ::8].cpu().numpy()-g["output_H_sub"]),99), np.abs(o["output_H"][..., ::8, ::8].cpu().numpy()-g["output_H_sub"]).max())
warp2 = o["output_H"][:, :3].contiguous()
tr=[]
with torch.no_grad():
    fo, lo = nets.flowformer(nets.W(sd,"flow_backbone."), a, warp2.cpu(), trace=tr)
tg=[]
fg, c1, _ = m.flow_backbone.flow_rows(a.cuda(), warp2, trace=tg)
def rows(y,H,W): return y.cpu().reshape(1,H,W,-1).permute(0,3,1,2)
print("context", (rows(tg[0]["context"],64,64)-tr[0]["context"]).abs().max().item(), tr[0]["context"].abs().max().item())
print("feat_s", (rows(tg[0]["feats"][0],64,64)-tr[0]["feat_s"]).abs().max().item())
print("feat_t", (rows(tg[0]["feats"][1],64,64)-tr[0]["feat_t"]).abs().max().item())
memg = (tg[0]["mem"]+tg[0]["short"]).cpu().view(4096,8,128)
print("mem", (memg-tr[0]["cost_memory"]).abs().max().item(), tr[0]["cost_memory"].abs().max().item())
for it in range(12):
    cg = tg[1+it]["coords1"].cpu().view(1,64,64,2).permute(0,3,1,2) - nets.coords_grid(1,64,64)
    # oracle trace lacks coords; use net
    print(it, "net", (rows(tg[1+it]["net"],64,64)-tr[1+it]["net"]).abs().max().item(), "cf", (rows(tg[1+it]["corr"][:, :81],64,64)-tr[1+it]["cost_forward"]).abs().max().item(), "cg", (rows(tg[1+it]["corr"][:, 84:],64,64)-tr[1+it]["cost_global"]).abs().max().item())
print("flow", (fg.cpu()-fo).abs().max().item())
