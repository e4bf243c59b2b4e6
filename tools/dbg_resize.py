import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import stitch_amd
ops = stitch_amd.ops
x = torch.rand(1, 3, 300, 340) * 255
s = 512 / 300
ref = F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False)
oh, ow = int(300 * s), int(340 * s)
print(ref.shape, oh, ow)
got = ops.resize_bilinear(x.cuda(), oh, ow, 2, div=(1.0 / s, 1.0 / s)).cpu()
print('align2 err', (got - ref).abs().max().item())
got0 = ops.resize_bilinear(x.cuda(), oh, ow, 0).cpu()
print('align0 (H/oh ratio) err', (got0 - ref).abs().max().item())
