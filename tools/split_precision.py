import torch, sys
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
torch.manual_seed(0)
for (M,N,K) in [(4096,256,1920),(8192,8192,512),(65536,128,128),(8192,256,1920),(1000,130,100)]:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda')/K**0.5; c0=torch.empty(M,N,device='cuda'); c1=torch.empty(M,N,device='cuda')
    ref=(a[:512].double()@w.double().t())
    ops.conv_gemm(a,w,c0,precision=0,split_k=1); ops.conv_gemm(a,w,c1,precision=1,split_k=1)
    e0=(c0[:512].double()-ref).abs().max().item(); e1=(c1[:512].double()-ref).abs().max().item()
    t0=timeit(lambda: ops.conv_gemm(a,w,c0,precision=0,split_k=1)); t1=timeit(lambda: ops.conv_gemm(a,w,c1,precision=1,split_k=1))
    print(f"M={M} N={N} K={K}: err fp32 {e0:.3e} split {e1:.3e} (|ref|max {ref.abs().max():.2f})  time fp32 {t0:.1f}us ({2*M*N*K/t0/1e6:.1f} TF) split {t1:.1f}us ({2*M*N*K/t1/1e6:.1f} TF)")
x=torch.randn(8192,384,device='cuda'); w=torch.randn(256,5*384,device='cuda')/44; c0=torch.empty(8192,256,device='cuda'); c1=torch.empty_like(c0)
for prec,c in ((0,c0),(1,c1)):
    ops.conv_gemm(x,w,c,geom=(2,64,64,1,5,1,1,0,2),precision=prec)
print("conv 1x5 max diff", (c0-c1).abs().max().item(), "time", timeit(lambda: ops.conv_gemm(x,w,c0,geom=(2,64,64,1,5,1,1,0,2),precision=0)), timeit(lambda: ops.conv_gemm(x,w,c1,geom=(2,64,64,1,5,1,1,0,2),precision=1)))
print("---- tiles")
for (M,N,K) in [(8192,256,1920),(8192,8192,512),(65536,256,1920),(65536,128,128)]:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda')/K**0.5; c=torch.empty(M,N,device='cuda')
    r=[]
    for tile in (3,2,1):
        t1=timeit(lambda: ops.conv_gemm(a,w,c,precision=1,split_k=1,tile=tile))
        r.append(f"t{tile}: {t1:.1f}us {2*M*N*K/t1/1e6:.1f}TF")
    print(f"M={M} N={N} K={K} split: "+"  ".join(r))
