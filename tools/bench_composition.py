"""Time the composition stage on one 512x544 canvas pair: HIP events around compose(), eager."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import stitch_amd
from oracle import composition as oc
net = stitch_amd.composition.Network(); net.load_state_dict(oc.seeded_state_dict(4321)); net = net.cuda().eval()
o1, o2, m1, m2 = (t.cuda() for t in oc.synthetic_inputs(512, 544, 77))
for _ in range(3): stitch_amd.composition.compose(net, o1, o2, m1, m2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): stitch_amd.composition.compose(net, o1, o2, m1, m2)
e1.record(); torch.cuda.synchronize()
gpu_ms = e0.elapsed_time(e1) / 10
sd = oc.seeded_state_dict(4321)
torch.set_num_threads(16)
c = [t.cpu() for t in (o1, o2, m1, m2)]
t0 = time.time(); oc.compose(sd, *c); cpu_s = time.time() - t0
# roofline line of the stage: FLOPs of its convolution GEMMs (library observer, as bench.py counts them) over their summed
# HIP-event durations against the 157.3 TFLOP/s fp32 matrix peak; the composition U-Net is small (its wide layers run at 128..32 px)
import bench as _bench
inst = _bench.instrumented_step(lambda: stitch_amd.composition.compose(net, o1, o2, m1, m2))
print(f"composition roofline: {inst['flops'] / 1e9:.1f} GFLOP in {inst['launches']} GEMM launches, {inst['ms']:.3f} ms of GEMM kernel time -> "
      f"{inst['flops'] / inst['ms'] / 1e9:.1f} TFLOP/s = {inst['flops'] / inst['ms'] / 1e9 / _bench.FP32_MFMA_PEAK_TFLOPS:.3f} of the fp32 MFMA peak; "
      f"whole stage {gpu_ms:.2f} ms -> {inst['flops'] / gpu_ms / 1e9 / _bench.FP32_MFMA_PEAK_TFLOPS:.3f} of the peak; algorithmic bytes {inst['alg_bytes'] / 1e6:.0f} MB "
      f"({inst['alg_bytes'] / gpu_ms / 1e6 / _bench.HBM_PEAK_GBS:.3f} of the HBM peak over the stage)")
print(f"composition 512x544: GPU {gpu_ms:.2f} ms / pair ({1e3 / gpu_ms:.1f} pairs/s); CPU oracle (16 threads) {cpu_s * 1e3:.0f} ms")
