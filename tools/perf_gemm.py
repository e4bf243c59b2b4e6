import torch, time, sys
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
    return s.elapsed_time(e)/n
for (M,N,K,tile) in [(4096,4096,256,0),(8192,8192,4096,1),(4096,128,2560,0),(4096,128,4096,0),(262144,128,128,0),(32768,512,128,0),(4096,256,2304,0)]:
    a=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda'); c=torch.empty(M,N,device='cuda')
    ms=timeit(lambda: ops.conv_gemm(a,w,c,tile=tile))
    print(f"gemm M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s  out-write {M*N*4/ms/1e6:.1f} GB/s")
f1=torch.randn(8,4096,256,device='cuda'); f2=torch.randn(8,4096,256,device='cuda'); vol=torch.empty(8,4096,4096,device='cuda')
ms=timeit(lambda: ops.corr_volume(f1,f2,vol))
print(f"corr B=8: {ms*1e3:.1f} us {8*8.59/ms:.1f} TFLOP/s hbm {8*75.5e6/ms/1e9:.2f} TB/s")
