import torch, sys
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for (Cin,Co,kh,kw,ph,pw) in [(384,256,1,5,0,2),(384,128,1,5,0,2),(128,256,3,3,1,1),(256,192,3,3,1,1),(128,64,3,3,1,1),(256,126,3,3,1,1),(128,384,1,5,0,2),(4,128,7,7,3,3)]:
    x=torch.randn(4096,Cin,device='cuda'); w=torch.randn(Co,kh*kw*Cin,device='cuda'); c=torch.empty(4096,Co,device='cuda')
    res=[]
    for split in (1,2,3,4,6,8,12,0):
        us=timeit(lambda: ops.conv_gemm(x,w,c,geom=(1,64,64,kh,kw,1,1,ph,pw),split_k=split))
        res.append(f"s{split}:{us:6.1f}")
    print(f"conv {kh}x{kw} {Cin}->{Co}: "+" ".join(res)+f"  ideal_mfma={2*4096*Co*kh*kw*Cin/157.3e6:6.1f}us")
for (M,N,K) in [(4096,128,4096),(4096,256,148),(4096,64,84),(4096,576,256),(128,4096,128)]:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); c=torch.empty(M,N,device='cuda')
    res=[]
    for split in (1,2,4,8,16,0):
        us=timeit(lambda: ops.conv_gemm(a,w,c,split_k=split))
        res.append(f"s{split}:{us:6.1f}")
    print(f"gemm {M}x{N}x{K}: "+" ".join(res)+f"  ideal_mfma={2*M*N*K/157.3e6:6.1f}us")
