"""Time the out.py stages after the forward on one 512x512 pair (HIP events): test_out forward, TPS post-pipeline with the
configured mix_fn (pass-through inpainter), composition.  Random-init weights, structured synthetic pair."""
import importlib, sys, time, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
from stitch_amd.data import structured_pair
cfg, tpc = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
from oracle import spec            # seeded weights (bounded scales): random init drives the flow past the pipeline's flow limit
model = stitch_amd.build_model(cfg)
model.load_state_dict(spec.seeded_state_dict(1234), strict=True)
model = model.cuda().eval()
comp = stitch_amd.composition.Network().cuda().eval()
a, b = (t.cuda() for t in structured_pair(512, 512, seed=7))
mix_fn = importlib.import_module(f"stitch_amd.mix_methods.{tpc.mix_method}").mix_fn
inp = importlib.import_module("stitch_amd.mix_methods.utils.passthrough_inpainter").inpainter
fn = lambda **kw: mix_fn(**kw, inpainter=inp, use_composition=False, is_plot=False, resize_to_area_limit_before_inpaint=750 * 750)
def run(tps_method):
    tpc.tps_method = tps_method
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter()
    ev[0].record()
    out = model(a, b, type="test_out")
    ev[1].record()
    inputs = dict(output1=out["output1"], mask1=out["mask1"], H_warp=out["H_warp"], H_warp_mask=out["H_warp_mask"], final_warp=out["final_warp"],
                  mask2=out["mask2"], residual_flow=out["residual_flow"], valid=None, occlusion_mask=out["occlusion_mask"],
                  border_points_mask=out["occlusion_mask"])
    limit = dict(width_min=out["width_min"], height_min=out["height_min"], out_height=out["out_height"], out_width=out["out_width"])
    new = stitch_amd.tps_pipeline.tps_H_warp(inputs, limit, tpc, inpaint_fn=fn)
    ev[2].record()
    m1, m2 = (out["mask1"] > 0.5).float(), (new["mask2"].repeat(1, 3, 1, 1) > 0.5).float()
    c = stitch_amd.composition.compose(comp, out["output1"], new["output2"], m1, m2)
    ev[3].record(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)], wall, new["points_src"].shape[1], (out["out_height"], out["out_width"])
for m in ("kornia", "opencv"):
    run(m)
    ts, wall, n, canvas = run(m)
    print(f"tps_method={m}: canvas {canvas}, {n} control points: forward {ts[0]:.2f} ms, post-pipeline {ts[1]:.2f} ms, composition {ts[2]:.2f} ms; wall {1e3*wall:.1f} ms")

# the network above leaves no control point on this pair; time the post-pipeline alone on a synthetic test_out-shaped case
from types import SimpleNamespace
from oracle import tps_pipeline as otp
case = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in otp.synthetic_case(5, 512, 544, -21, -13, 548, 588).items()}
limit = dict(width_min=-21, height_min=-13, out_height=548, out_width=588)
for m in ("kornia", "opencv"):
    tpc.tps_method = m
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        new = stitch_amd.tps_pipeline.tps_H_warp(case, SimpleNamespace(**limit), tpc, inpaint_fn=fn)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pts_a, pts_b = new["points_dst"][0].float().cuda(), new["points_src"][0].float().cuda()
    e0.record(); stitch_amd.ops.tps2_solve(pts_a, pts_b if m == "kornia" else pts_a, pts_b, mode=0 if m == "kornia" else 1); e1.record(); torch.cuda.synchronize()
    print(f"synthetic 548x588 canvas, tps_method={m}: {new['points_src'].shape[1]} control points, tps_H_warp + mix_fn wall {1e3*dt:.2f} ms (TPS solve alone {e0.elapsed_time(e1):.2f} ms)")
