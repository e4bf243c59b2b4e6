import torch, sys
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for (M,N,K) in [(8192,8192,4096),(8192,8192,512),(8192,256,1920),(16384,256,1920),(65536,256,1920),(65536,128,128),(65536,128,2048),(8192,8192,128)]:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); c=torch.empty(M,N,device='cuda')
    r=[]
    for tile in (1,2,3):
        us=timeit(lambda: ops.conv_gemm(a,w,c,tile=tile,split_k=1))
        r.append(f"t{tile}: {us:9.1f}us {2*M*N*K/us/1e6:6.1f}TF")
    print(f"M={M} N={N} K={K}: "+"  ".join(r))
