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
shapes=[(32768,128,128),(32768,512,128),(32768,128,512),(32768,384,128),(4096,64,64),(4096,128,128),(262144,128,128),(262144,256,128),(262144,128,64),(4096,256,148),(8192,1024,256),(16384,512,128),(16384,128,512)]
for (M,N,K) in shapes:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); c=torch.empty(M,N,device='cuda')
    res=[]
    for tile in (1,2,3,4):
        us=timeit(lambda: ops.conv_gemm(a,w,c,tile=tile,split_k=1))
        res.append(f"t{tile}:{us:7.1f}us")
    us=timeit(lambda: ops.conv_gemm(a,w,c))
    print(f"M={M:>7} N={N:>5} K={K:>5} "+" ".join(res)+f" auto:{us:7.1f}us  ideal_hbm={(M*K+M*N+N*K)*4/5e6:6.1f}us ideal_mfma={2*M*N*K/157.3e6:6.1f}us")
# conv shapes (decoder)
for (Cin,Co,kh,kw,ph,pw) in [(512,128,1,5,0,2),(128,256,3,3,1,1),(256,192,3,3,1,1),(256,2,3,3,1,1),(256,126,3,3,1,1)]:
    x=torch.randn(4096,Cin,device='cuda'); w=torch.randn(Co,kh*kw*Cin,device='cuda'); c=torch.empty(4096,Co,device='cuda')
    res=[]
    for split in (1,2,4,8,0):
        us=timeit(lambda: ops.conv_gemm(x,w,c,geom=(1,64,64,kh,kw,1,1,ph,pw),split_k=split))
        res.append(f"s{split}:{us:7.1f}us")
    print(f"conv {kh}x{kw} {Cin}->{Co}: "+" ".join(res)+f"  ideal_mfma={2*4096*Co*kh*kw*Cin/157.3e6:6.1f}us")
