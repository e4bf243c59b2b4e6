"""Drop-in ``FlowHomoAdpater`` (reference: core/flowHomoAdpater.py:38-377) on the MI355X kernels.

Same constructor, same ``forward(input1_tensor, input2_tensor, type, pad_mode, preprocess_callback)``,
same output-dict keys / shapes / dtypes, same checkpoint key set (``homo_backbone.*``,
``flow_backbone.*``; a DataParallel ``module.`` prefix is accepted).  Inputs are float32
``[B,3,H,W]`` RGB in 0..255.  Everything below runs in HIP kernels through the C-ABI; PyTorch only
owns the device buffers.  The live configuration is the shipped one (configs/last_config.py +
inf_configs/*): only_homo=False, use_forward=False, use_combine_h_flow=False,
use_fb_consistency_mask=True, test_not_use_combine_h_flow=True, use_whole_resolution=False; the
reference's other branches raise ``NotImplementedError`` here exactly where they are dead/buggy there.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops


def preprocess_occlusion_mask(occlusion_mask, kernel_size=(19, 19)):
    """reference: core/flowHomoAdpater.py:18-35 (threshold, 19x19 morphological open)."""
    if kernel_size[0] != kernel_size[1]:
        raise NotImplementedError("square kernels only")
    return ops.morph_open(occlusion_mask.contiguous(), kernel_size[0])


def _flag(cfg, name, default=False):
    return getattr(cfg, name) if hasattr(cfg, name) else default


class FlowHomoAdpater(nn.Module):
    def __init__(self, homo_backbone, flow_backbone, cfg):
        super().__init__()
        self.cfg = cfg
        self.use_forward = _flag(cfg, "use_forward")
        self.detach_H = _flag(cfg, "detach_H")
        self.detach_flow = _flag(cfg, "detach_flow")
        self.homo_backbone = homo_backbone
        self.flow_backbone = flow_backbone
        self._host = {}
        self._eval_pipeline = None       # stitch_amd.evaluate.EvalPipeline: hipGraphs of the evaluation loop, captured on first use

    # checkpoints saved from DataParallel carry a "module." prefix (out.py:80-85)
    def load_state_dict(self, state_dict, strict=True, assign=False):
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        self._eval_pipeline = None       # captured graphs point at the previous weights' packed copies
        return super().load_state_dict(state_dict, strict=strict)

    def weights_generation(self, deep=False):
        """Identity of the weights a captured hipGraph was recorded against.  The backbones drop their packed copies (and bump ``_gen``) on
        ``load_state_dict`` and on every ``_apply`` (``.cuda()`` / ``.float()`` / ``.to()``), also when called on a sub-module alone
        (``model.flow_backbone.load_state_dict(...)``): a graph recorded before that would replay kernels that read freed or stale packed
        copies next to live parameters.  Every graph holder (``GraphedForward``, ``GraphedTestOut``, ``evaluate.EvalPipeline``) stores this
        value at capture and re-captures on mismatch.  ``deep=True`` adds the parameters' in-place version counters (``p.data`` writes,
        optimiser steps) and storage addresses -- 699 tensors, checked once per harness run rather than per pair."""
        if deep:
            v = a = 0
            for t in self.parameters():
                v += t._version
                a ^= t.data_ptr()
            seen = getattr(self, "_deep_seen", None)
            if seen != (v, a):
                # an in-place write (or the first deep check after the backbones packed): the PACKED copies (BatchNorm folded in, fused q|k|v,
                # split3 planes) are stale too -- drop them, which also bumps the generation the shallow holders compare
                if seen is not None or getattr(self.homo_backbone, "_pk", None) is not None or getattr(self.flow_backbone, "_pk", None) is not None:
                    self.homo_backbone._invalidate()
                    self.flow_backbone._invalidate()
                self._deep_seen = (v, a)
            return (getattr(self.homo_backbone, "_gen", 0), getattr(self.flow_backbone, "_gen", 0), v, a)
        return (getattr(self.homo_backbone, "_gen", 0), getattr(self.flow_backbone, "_gen", 0))

    # ------------------------------------------------------------------ small host-side constants
    def _mat(self, dev, key, rows):
        k = (key, str(dev))
        if k not in self._host:
            self._host[k] = torch.tensor(rows, dtype=torch.float32).to(dev)
        return self._host[k]

    def _scale_pair(self, dev, w, h):
        """M = [[w/2,0,w/2],[0,h/2,h/2],[0,0,1]] and its inverse (flowHomoAdpater.py:98-107)."""
        k = ("scale", float(w), float(h), str(dev))
        if k not in self._host:
            M = torch.tensor([[w / 2.0, 0., w / 2.0], [0., h / 2.0, h / 2.0], [0., 0., 1.]], dtype=torch.float32).to(dev)
            self._host[k] = (M, self._inv3(M))
        return self._host[k]

    def _inv3(self, M):
        """torch.inverse of a 3x3 in the reference's arithmetic, on the device: eye @ inverse(M) @ eye is exact."""
        eye = self._mat(M.device, "eye", [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]])
        out = torch.empty((1, 3, 3), device=M.device)
        ops.mat3_sandwich(eye, M.reshape(1, 3, 3).contiguous(), eye, out, invert=True)
        return out[0]

    def _corners(self, dev, w, h):
        return self._mat(dev, ("corners", float(w), float(h)), [[0., 0.], [w, 0.], [0., h], [w, h]])

    # ------------------------------------------------------------------ reference surface
    def predict_homo(self, input1_tensor, input2_tensor):
        """[0,255] -> corner offsets [B,4,2] (flowHomoAdpater.py:53-61); x/127.5 - 1 is fused into the prep."""
        off = self.homo_backbone.offsets_from_images(input1_tensor, input2_tensor, 1.0, 127.5, 1.0)
        return off.reshape(-1, 4, 2)

    def predict_flow(self, input1_tensor, input2_tensor):
        """flow 1->2 at full resolution, eval: list of one tensor (flowHomoAdpater.py:63-70)."""
        return [self.flow_backbone.flow_rows(input1_tensor, input2_tensor)[0]]

    def predict_flow_pair(self, input1_tensor, input2_tensor):
        """(flow 1->2, flow 2->1) from one batched FlowFormer evaluation (see FlowFormer.flow_rows_pair)."""
        B = input1_tensor.shape[0]
        both = self.flow_backbone.flow_rows_pair(input1_tensor, input2_tensor)[0]
        return both[:B], both[B:]

    def forward(self, input1_tensor, input2_tensor, type="train", pad_mode="constant", preprocess_callback=None):
        # The reference's callers go through nn.DataParallel and hand over whatever the loader produced, CPU tensors included (out.py:197,
        # evaluate.py:43): the inputs move to the module's device, as DataParallel's scatter does.  The COMPUTE has no CPU path: a module
        # that itself sits on the CPU still raises.
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("FlowHomoAdpater (gfx950) runs on the MI355X HIP kernels only: move the module to cuda (there is no CPU fallback)")
        input1_tensor = input1_tensor.to(dev, non_blocking=True).float().contiguous()
        input2_tensor = input2_tensor.to(dev, non_blocking=True).float().contiguous()
        with torch.no_grad():
            if type == "test_out":
                return self.test_out_forward(input1_tensor, input2_tensor, pad_mode=pad_mode,
                                             preprocess_callback=preprocess_callback)
            if type == "test_eval":
                if self.training:
                    raise NotImplementedError("inference-only drop-in: call .eval() first (the training branch of "
                                              "train_eval_foward, flowHomoAdpater.py:83-191, is not implemented)")
                return self.train_eval_foward(input1_tensor, input2_tensor)
            if type == "train":
                # the reference returns every refinement prediction with gradients here; this path is no_grad with
                # frozen parameters, so training through it would silently learn nothing
                raise NotImplementedError("type='train' is not supported by the gfx950 inference path")
            raise NotImplementedError

    def graphed(self, type="test_eval"):
        """hipGraph replay of ``forward(type=...)`` for fixed-shape inputs (see ``GraphedForward``)."""
        return GraphedForward(self, type)

    # ------------------------------------------------------------------ eval @ fixed size (:83-191)
    def train_eval_foward(self, input1_tensor, input2_tensor):
        if self.use_forward or _flag(self.cfg, "use_combine_h_flow") or _flag(self.cfg, "only_homo"):
            raise NotImplementedError("only the shipped branch (flowHomoAdpater.py:165-186) is implemented")
        dev = input1_tensor.device
        B, _, img_h, img_w = input1_tensor.shape
        motion = self.predict_homo(input1_tensor, input2_tensor)
        H = torch.empty((B, 3, 3), device=dev)
        ops.dlt4(self._corners(dev, img_w, img_h), motion.contiguous(), H, B, 1.0, 1.0, 8.0)          # :96
        M, Minv = self._scale_pair(dev, img_w / 8, img_h / 8)
        H_mat, H_inv_mat = torch.empty_like(H), torch.empty_like(H)
        ops.mat3_sandwich(Minv, H, M, H_mat)                                                           # :108
        ops.mat3_sandwich(Minv, H, M, H_inv_mat, invert=True)                                          # :112
        output_H = ops.homo_warp(input2_tensor, H_mat.view(B, 9), (img_h, img_w), n_ones=3)            # :111
        output_H_inv = ops.homo_warp(input1_tensor, H_inv_mat.view(B, 9), (img_h, img_w), n_ones=3)    # :113
        warp2 = output_H[:, 0:3].contiguous()
        fb = _flag(self.cfg, "use_fb_consistency_mask")              # missing key = False, as hasattr(...) and ... (:176)
        if fb:
            flow_ij, flow_ji = self.predict_flow_pair(input1_tensor, warp2)                            # :167 and :178, one batch
        else:
            flow_ij = self.predict_flow(input1_tensor, warp2)[0]                                       # :167
        final = ops.flow_warp(output_H, flow_ij)                                                       # :170
        out = dict()
        if fb:
            occ = ops.occlusion_from_range(ops.range_map(flow_ji.contiguous()), True)                  # :180-181
        else:
            occ = torch.ones((B, 1, img_h, img_w), device=dev)
        overlap = ops.eval_finish(final, occ)                                                          # :171-174,182
        if fb:
            out.update(origin_occlusion_mask=occ.squeeze(1))
        out.update(output_H=output_H, output_H_inv=output_H_inv, final_warp_output=final, overlap=overlap,
                   flow_predictions=[flow_ij], H=H)
        return out

    # ------------------------------------------------------------------ stitching @ native size (:197-377)
    def test_out_forward(self, input1_tensor, input2_tensor, pad_mode="constant", preprocess_callback=None):
        nets = self._test_out_nets(input1_tensor, input2_tensor)
        return self._test_out_canvas(input1_tensor, input2_tensor, nets)

    def _test_out_nets(self, input1_tensor, input2_tensor):
        """First part of test_out_forward (:204-266): both networks at 512x512, native-resolution DLT and the mesh bounds.
        No host synchronisation and a fixed launch sequence for a given input shape, so it can be replayed from a hipGraph
        (``GraphedTestOut``); everything it returns is a device tensor."""
        if self.use_forward:
            raise NotImplementedError
        if not _flag(self.cfg, "test_not_use_combine_h_flow") or _flag(self.cfg, "use_whole_resolution"):
            raise NotImplementedError("only the shipped branch (test_not_use_combine_h_flow=True, use_whole_resolution=False, "
                                      "flowHomoAdpater.py:303-360) is implemented")
        if not _flag(self.cfg, "use_fb_consistency_mask"):
            raise NotImplementedError("shipped inference config sets use_fb_consistency_mask=True (flowHomoAdpater.py:324)")
        dev = input1_tensor.device
        B, _, img_h, img_w = input1_tensor.shape
        if B != 1:
            raise NotImplementedError("test_out shares one data-dependent canvas: batch must be 1 (as in out.py:37)")
        pre_out = None
        if os.environ.get("ST_EXP_PREALLOC") == "1":           # experiment: the two flow outputs do not reuse blocks the networks freed
            pre_out = (torch.empty((B, 2, img_h, img_w), device=dev), torch.empty((B, 2, img_h, img_w), device=dev))
        a512 = ops.resize_bilinear(input1_tensor, 512, 512, False)                                     # :204-205
        b512 = ops.resize_bilinear(input2_tensor, 512, 512, False)
        motion = self.predict_homo(a512, b512).contiguous()
        H512 = torch.empty((B, 3, 3), device=dev)
        ops.dlt4(self._corners(dev, 512., 512.), motion, H512, B, 1.0, 1.0, 1.0)                       # :216
        M5, M5inv = self._scale_pair(dev, 512., 512.)
        th = torch.empty_like(H512)
        ops.mat3_sandwich(M5inv, H512, M5, th)
        out_H = ops.homo_warp(b512, th.view(B, 9), (512, 512), n_ones=3)                               # :230
        warp2_512 = out_H[:, 0:3].contiguous()
        warp_mask_512 = ops.mean_threshold(out_H[:, 3:6].contiguous(), 0.5)                            # :233-234
        flow512, back512 = self.predict_flow_pair(a512, warp2_512)                                     # :236 and :326, one batch
        residual = ops.resize_bilinear(flow512, img_h, img_w, True, div=(512 / float(img_w), 512 / float(img_h)), out=pre_out[0] if pre_out else None)  # :241
        back = ops.resize_bilinear(back512, img_h, img_w, True, div=(512 / float(img_w), 512 / float(img_h)), out=pre_out[1] if pre_out else None)
        H = torch.empty((B, 3, 3), device=dev)
        ops.dlt4(self._corners(dev, float(img_w), float(img_h)), motion, H, B, img_w / 512.0, img_h / 512.0, 1.0)  # :244-253
        bounds = torch.empty((4,), device=dev)
        ops.mesh_bounds(H, bounds, img_w, img_h)                                                       # :254-266
        if os.environ.get("ST_EXP_KEEP") == "1":                  # diagnostics (tools/graph_race_stress.py): the 512 x 512 flows as extra static outputs
            return dict(residual=residual, back=back, H=H, bounds=bounds, warp2_512=warp2_512, warp_mask_512=warp_mask_512, flow512=flow512, back512=back512)
        return dict(residual=residual, back=back, H=H, bounds=bounds, warp2_512=warp2_512, warp_mask_512=warp_mask_512)

    def _test_out_canvas(self, input1_tensor, input2_tensor, nets):
        """Second part (:268-377): the canvas size is read back to the host (the path's one sync), then the canvas-sized
        homography warps, the flow warp, occlusion mask and the blend."""
        dev = input1_tensor.device
        B, _, img_h, img_w = input1_tensor.shape
        residual, back, H = nets["residual"], nets["back"], nets["H"]
        mnx, mxx, mny, mxy = nets["bounds"].tolist()                                                   # the path's host sync (:268,367)
        width_max, width_min = int(max(float(img_w), mxx)), int(min(0.0, mnx))                         # .int() truncation
        height_max, height_min = int(max(float(img_h), mxy)), int(min(0.0, mny))
        out_width, out_height = width_max - width_min, height_max - height_min                         # :270-271
        Mt = torch.tensor([[out_width / 2.0, 0., out_width / 2.0], [0., out_height / 2.0, out_height / 2.0], [0., 0., 1.]]).to(dev)
        _, Ninv = self._scale_pair(dev, float(img_w), float(img_h))                                   # :274-276
        I_ = torch.tensor([[1., 0., float(width_min)], [0., 1., float(height_min)], [0., 0., 1.]]).to(dev)
        I_mat = torch.empty((1, 3, 3), device=dev)
        ops.mat3_sandwich(Ninv, I_.view(1, 3, 3), Mt, I_mat)                                          # :291
        canvas = (out_height, out_width)
        homo_output = ops.homo_warp(input1_tensor, I_mat.view(1, 9), canvas, n_ones=3)                 # :292
        ident = self._mat(dev, "eye", [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]])
        Hc = torch.empty_like(H)
        ops.mat3_sandwich(ident, H, I_, Hc)                                                    # H @ I_  (:306)
        H_mat = torch.empty_like(H)
        ops.mat3_sandwich(Ninv, Hc, Mt, H_mat)                                         # :307
        homo_output2 = ops.homo_warp(input2_tensor, H_mat.view(B, 9), canvas, n_ones=3)                # :310
        rf = ops.homo_warp(residual, I_mat.view(1, 9), canvas, n_ones=1)                               # :313-314
        final = ops.flow_warp(homo_output2, rf[:, 0:2].contiguous(), rf[:, 2:3].contiguous())          # :316-317
        occ = ops.occlusion_from_range(ops.range_map(back), False)                                     # :332
        origin_occ = ops.morph_open(occ, 19)                                                           # :333-334
        occ_c = ops.homo_warp(origin_occ, I_mat.view(1, 9), canvas)                                    # :335
        occ_c = ops.morph_open(occ_c, 19)                                                              # :336
        output2, mask1, mask2, blend = ops.blend(homo_output, homo_output2, final, occ_c)              # :339-360
        return dict(H_warp=homo_output2[:, 0:3], final_warp=final[:, 0:3], output1=homo_output[:, 0:3], output2=output2,
                    mask1=mask1, mask2=mask2, blend_image=blend, residual_flow=residual, width_min=width_min,
                    height_min=height_min, out_height=out_height, out_width=out_width, H=Hc,
                    warp_input2_mask=nets["warp_mask_512"], warp_input2_tensor_512=nets["warp2_512"], I_mat=I_mat,
                    H_warp_mask=homo_output2[:, 3:6], occlusion_mask=occ_c, origin_occlusion_mask=origin_occ)

    def graphed_test_out(self):
        """hipGraph replay of the network part of ``forward(type="test_out")`` (see ``GraphedTestOut``)."""
        return GraphedTestOut(self)


class GraphedTestOut:
    """``forward(type="test_out")`` with its network part (both nets at 512x512, DLT, mesh bounds: ~1 000 launches, no host
    sync) replayed from a hipGraph per input shape; the canvas part, whose shapes depend on the read-back bounds, stays
    eager.  ``residual_flow`` / ``warp_input2_*`` live in the graph's static buffers: consume them before the next call."""

    def __init__(self, model):
        self.model = model
        self._graphs = {}

    def __call__(self, input1_tensor, input2_tensor):
        return self.finish(self.launch(input1_tensor, input2_tensor))

    def launch(self, input1_tensor, input2_tensor):
        """Enqueue the network part on the current stream (asynchronous) and return a handle for ``finish``.  Several
        GraphedTestOut objects on several streams can be launched before any of them is finished, so that the host-side
        wait of one pair (the canvas bounds) overlaps with the other pairs' kernels."""
        m = self.model
        if m.training:
            raise NotImplementedError("inference-only drop-in: call .eval() first")
        key = (tuple(input1_tensor.shape), input1_tensor.device.index)
        ent = self._graphs.get(key)
        gen = m.weights_generation()
        if ent is not None and ent[5] != gen:          # weights re-packed since the capture: the graph reads dead copies
            ent = None
            self._graphs.clear()
        with torch.no_grad():
            if ent is None:
                a, b = input1_tensor.float().contiguous().clone(), input2_tensor.float().contiguous().clone()
                ws = ops.new_workspace(a.device)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), ops.workspace_scope(ws):
                    for _ in range(2):
                        m._test_out_nets(a, b)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph), ops.workspace_scope(ws):
                    nets = m._test_out_nets(a, b)
                ent = self._graphs[key] = (graph, a, b, nets, ws, m.weights_generation())
            graph, a, b, nets = ent[:4]
            a.copy_(input1_tensor)
            b.copy_(input2_tensor)
            graph.replay()
        return (a, b, nets, torch.cuda.current_stream())

    def finish(self, handle):
        a, b, nets, stream = handle
        with torch.no_grad(), torch.cuda.stream(stream):
            return self.model._test_out_canvas(a, b, nets)


class GraphedForward:
    """Capture ``FlowHomoAdpater.forward(type="test_eval")`` into a hipGraph and replay it.

    One pair is ~1 900 kernel launches of a few microseconds each; replaying them from a captured
    graph removes the per-launch host cost (Python + ctypes + hipLaunchKernel).  The captured forward
    has no host synchronisation and a fixed launch sequence (12 refinement iterations), so the graph
    is exact.  Outputs live in the graph's static buffers: consume (or clone) them before the next call.
    ``test_out`` is not capturable as a whole (its canvas size is read back to the host mid-way).
    """

    def __init__(self, model, type="test_eval"):
        if type != "test_eval":
            raise NotImplementedError("only the fixed-shape test_eval path can be captured")
        self.model, self.type = model, type
        self._graphs = {}

    def __call__(self, input1_tensor, input2_tensor):
        key = (tuple(input1_tensor.shape), input1_tensor.device.index)
        ent = self._graphs.get(key)
        if ent is not None and ent[5] != self.model.weights_generation():       # weights re-packed since the capture
            ent = None
            self._graphs.clear()
        if ent is None:
            a, b = input1_tensor.float().contiguous().clone(), input2_tensor.float().contiguous().clone()
            ws = ops.new_workspace(a.device)            # this graph's own split-K slabs (graphs may replay concurrently)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), ops.workspace_scope(ws):
                for _ in range(2):                      # warm-up: weight prepack, constant tables, LDS attributes
                    self.model(a, b, type=self.type)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), ops.workspace_scope(ws):
                out = self.model(a, b, type=self.type)
            ent = self._graphs[key] = (graph, a, b, out, ws, self.model.weights_generation())
        graph, a, b, out = ent[:4]
        a.copy_(input1_tensor)
        b.copy_(input2_tensor)
        graph.replay()
        return out
