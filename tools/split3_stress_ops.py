"""Concurrency stress of the split3 OPERATORS (3 streams at once, repeated), each result compared bit for bit with the same call run alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stitch_amd
ops = stitch_amd.ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B, H, W = 2, 64, 64
N = H * W
R = B * N
hx = torch.randn(R, 384, generator=g).to(dev); hx[:, :128] = hx[:, :128].tanh()
tabs = [(torch.randn(R, 384, generator=g) * 0.3).to(dev) for _ in range(2)]
wzr = [ops.split3_pack((torch.randn(256, 5 * 384, generator=g) * 0.02).to(dev)) for _ in range(2)]
wq = [ops.split3_pack((torch.randn(128, 5 * 384, generator=g) * 0.02).to(dev)) for _ in range(2)]
attn = torch.softmax(torch.randn(B, N, N, generator=g).to(dev) * 2, -1)
attn_p = ops.split3_pack(attn.view(B * N, N))
w_v, gamma = (torch.randn(128, 128, generator=g) / 11.3).to(dev), torch.tensor([0.37]).to(dev)
cor1, flo1 = torch.randn(R, 256, generator=g).to(dev), torch.randn(R, 128, generator=g).to(dev)
cor1_p, flo1_p = ops.split3_pack(cor1), ops.split3_pack(flo1)
wc2, wf2 = ops.split3_pack((torch.randn(192, 9 * 256, generator=g) / 48).to(dev)), ops.split3_pack((torch.randn(64, 9 * 128, generator=g) / 34).to(dev))
g3 = (B, H, W, 3, 3, 1, 1, 1, 1)
corr = torch.randn(R, 160, generator=g).to(dev); wc1 = (torch.randn(256, 160, generator=g) / 12).to(dev)
coords = (torch.randn(R, 2, generator=g) * 3 + 20).to(dev); w98 = (torch.randn(98, 128, generator=g) * 0.1).to(dev); b98 = (torch.randn(128, generator=g) * 0.1).to(dev)

def run(st):
    """one of everything; st = dict of this stream's private buffers"""
    hA = st["hA"]; hA.copy_(hx)
    pA, pB = st["pA"], st["pB"]
    ops.split3_pack(hA, out=pA)
    ops.sepconv_gru_split3(hA, pA, pB, st["zb"], tabs[0], tabs[1], wzr[0], wq[0], wzr[1], wq[1], B, H, W)
    ops.gma_aggregate_split3(attn_p, hA[:, 128:256], w_v, gamma, st["vT"], st["vTp"], hA[:, 256:], pA.cols(256, 384), B, N)
    ops.conv_gemm_pair((cor1_p, wc2, st["cf"][:, :192], dict(geom=g3, act="relu", out_planes=st["cfp"].cols(0, 192), no_f32=True)),
                       (flo1_p, wf2, st["cf"][:, 192:], dict(geom=g3, act="relu", out_planes=st["cfp"].cols(192, 256), no_f32=True)))
    ops.conv_gemm(corr, wc1, st["c1"], act="relu", out_planes=st["c1p"])
    ops.flow_encode_split3(coords, w98, b98, st["f1"], hA[:, 254:256], B, H, W, st["f1p"], (pA, 254))

def bufs():
    return dict(hA=torch.empty(R, 384, device=dev), pA=ops.Planes(R, 384, dev), pB=ops.Planes(R, 384, dev), zb=torch.empty(R, 128, device=dev),
                vT=torch.empty(B, 128, N, device=dev), vTp=ops.Planes(B * 128, N, dev), cf=torch.zeros(R, 256, device=dev), cfp=ops.Planes(R, 256, dev),
                c1=torch.empty(R, 256, device=dev), c1p=ops.Planes(R, 256, dev), f1=torch.empty(R, 128, device=dev), f1p=ops.Planes(R, 128, dev))

def snap(st):
    return {k: (v.t[:, :4].clone() if k == "pB" else v.t.clone()) if isinstance(v, ops.Planes) else v.clone() for k, v in st.items()}

ref_st = bufs(); run(ref_st); torch.cuda.synchronize(); ref = snap(ref_st)
streams = [torch.cuda.Stream() for _ in range(3)]
sts = [bufs() for _ in range(3)]
wss = [ops.new_workspace(dev) for _ in range(3)]
bad = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    for s in range(3):
        with torch.cuda.stream(streams[s]), ops.workspace_scope(wss[s]):
            run(sts[s])
    torch.cuda.synchronize()
    for s in range(3):
        cur = snap(sts[s])
        for k in ref:
            if not torch.equal(cur[k], ref[k]):
                bad[k] = bad.get(k, 0) + 1
                if bad[k] <= 2:
                    print(f"iter {it} stream {s} {k}: {int((cur[k] != ref[k]).sum())} elements differ", flush=True)
print("mismatches:", bad)
