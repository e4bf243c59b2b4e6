import torch, sys
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
x=torch.randn(4096,384,device='cuda'); w=torch.randn(256,5*384,device='cuda'); c=torch.empty(4096,256,device='cuda')
for _ in range(3): ops.conv_gemm(x,w,c,geom=(1,64,64,1,5,1,1,0,2),split_k=1)
for _ in range(3): ops.conv_gemm(x,w,c,geom=(1,64,64,1,5,1,1,0,2),split_k=4)
a=torch.randn(8192,4096,device='cuda'); b=torch.randn(8192,4096,device='cuda'); o=torch.empty(8192,8192,device='cuda')
for _ in range(2): ops.conv_gemm(a,b,o,tile=1,split_k=1)
for _ in range(2): ops.conv_gemm(a,b,o,tile=3,split_k=1)
torch.cuda.synchronize()
