"""The 12 refinement iterations of one pair's two passes (decoder.py:289-344: cost lookup, token chain, update block) replayed from a hipGraph
on synthetic state: ms per 12 iterations and a per-kernel table of one eager iteration.    [ST_SPLIT3=0|1] python tools/decoder_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
from stitch_amd import flowformer as ff
ops = stitch_amd.ops
dev = torch.device("cuda:0")
cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
torch.manual_seed(1234)
m = stitch_amd.build_model(cfg).cuda().eval()
fb = m.flow_backbone
D = fb.pack()["dec"]
B, H1, W1 = 2, 64, 64
N = H1 * W1; R = B * N
g = torch.Generator().manual_seed(2)
ctx = torch.randn(R, 256, generator=g).to(dev)
cost_maps = torch.randn(R, N, generator=g).to(dev)
nl = fb._pk["latents"].shape[0]
kv = torch.randn(R * nl, 128, generator=g).to(dev)
with torch.no_grad():
    pre = fb._decoder_prologue(ctx, B, H1, W1)
    S, gru_tab, attn = pre["S"], pre["gru_tab"], pre["attn"]
    coords1 = torch.empty(R, 2, device=dev)

    def iters(n=12):
        ops.coords_grid(coords1, B, H1, W1)
        for _ in range(n):
            ops.cost_lookup9x9(cost_maps, coords1, S["corr"], R, H1, W1)
            ops.decoder_token_chain(S["corr"], coords1, kv, D["chain16"], R, nl)
            fb._update_block(S, coords1, attn, gru_tab, B, H1, W1)
    ws = ops.new_workspace(dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), ops.workspace_scope(ws):
        iters(2)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr), ops.workspace_scope(ws):
        iters(12)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        gr.replay()
    b.record(); torch.cuda.synchronize()
    print(f"SPLIT3={int(ff.SPLIT3)}: 12 iterations (both passes of a pair) = {a.elapsed_time(b) / 30:.3f} ms per replay, one stream")
