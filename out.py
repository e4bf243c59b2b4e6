"""Inference + save harness of the stitching path (re-statement of the reference's out.py:15-54,106-146,158-312,
SURVEY.md section 8 f-1 / f-3) on the MI355X package.

    python out.py --data_root_path ./demo/ --inf_cfg all_img1_with_inpaint_g12_transRef [--ckpt_path CKPT]

Same flags, `demo.txt` pair list (one directory per line holding input1.jpg / input2.jpg), RGB-float loading and
result-directory naming as the reference.  The forward (`type="test_out"`) and the TPS post-pipeline
(core/inference/tps_pipline.py, `stitch_amd.tps_pipeline`) run on the HIP kernels; the composition stage (out.py:277-312,
`cfg.use_composition`) runs on the post-TPS canvases.  The `mix_fn` plug-in named by `TPS_PIPELINE_CONFIG.mix_method` runs too
(`stitch_amd.mix_methods`).  Differences from the reference's files: the neural inpainter inside `mix_fn` (TransRef /
diffusion: fetched weights + third-party CUDA ops) is out of scope and replaced by a pass-through, so in `warp2.jpg`,
`mask2.jpg`, `ave_fusion.jpg`, `composition.jpg`, `learned_mask*.jpg` the thin border `mix_fn` leaves to the inpainter is not
synthesised; with the shipped `tps_method="opencv"` the spline is this package's own pixel-unit TPS (OpenCV is not
installable here: unpinned against OpenCV)."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def get_config(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--data_root_path", type=str, default="./demo/")
    p.add_argument("--txt_file", type=str, default="demo.txt")
    p.add_argument("--result_dir", type=str, default="results")
    p.add_argument("--ckpt_path", "--restore_ckpt", dest="restore_ckpt", type=str, default="",
                   help="checkpoint (reference flag: --ckpt_path, out.py:18; --restore_ckpt kept as an alias); empty = random init")
    p.add_argument("--model_config_name", type=str, default="last_config")
    p.add_argument("--inf_cfg", type=str, default="all_img1_with_inpaint_g12_transRef")
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--skip_if_avg_fusion_exists", action="store_true")
    p.add_argument("--dry-run", dest="dry_run", action="store_true",
                   help="list the pairs this rank would process (sharding rehearsal: no model, no GPU) and exit")
    args = p.parse_args(argv)
    import stitch_amd
    cfg, tps = stitch_amd.load_inference_config(args.inf_cfg, args.model_config_name)
    for k, v in vars(args).items():
        if k == "restore_ckpt" and not v:
            continue
        cfg[k] = v
    cfg.TPS_PIPELINE_CONFIG = tps
    return cfg


def get_data_dict_list(data_root_path, txt_file):
    """out.py:106-123: one pair directory per line of the list file."""
    out = []
    with open(os.path.join(data_root_path, txt_file)) as f:
        for line in f:
            if line.strip():
                out.append({"DATA_PATH": os.path.join(data_root_path, line.strip()), "IMG1": "input1.jpg", "IMG2": "input2.jpg"})
    return out


def decodeSingleData(data_path, img1_name, img2_name):
    """host half of out.py:129-136 (cv2.imread + BGR->RGB there; PIL RGB decode here): two uint8 [H,W,3] arrays.  Runs on a
    prefetch thread one pair ahead of the GPU (PIL releases the GIL while decoding)."""
    from PIL import Image
    return tuple(np.ascontiguousarray(np.array(Image.open(os.path.join(data_path, name)).convert("RGB")).astype(np.uint8)[..., :3])
                 for name in (img1_name, img2_name))


def uploadSingleData(arrays, resize_to_512=False):
    """device half of out.py:137-143: uint8 HWC -> float [1,3,H,W] in 0..255 on the GPU (`st_load_rgb8`, exact), optional 512 resize."""
    import stitch_amd
    ts = []
    for arr in arrays:
        u8 = torch.from_numpy(arr).unsqueeze(0).cuda(non_blocking=True)
        hw = arr.shape[0] * arr.shape[1]
        t = stitch_amd.ops.load_rgb8(u8) if hw % 4 == 0 else u8.permute(0, 3, 1, 2).float().contiguous()
        ts.append(stitch_amd.ops.resize_bilinear(t, 512, 512, False) if resize_to_512 else t)
    return ts[0], ts[1]


def loadSingleData(data_path, img1_name, img2_name, resize_to_512=False):
    """out.py:129-146 -> two float [1,3,H,W] tensors, 0..255 (host tensors, as the reference returns them)."""
    a, b = uploadSingleData(decodeSingleData(data_path, img1_name, img2_name), resize_to_512)
    return a.cpu(), b.cpu()


def to_pillow(t):
    from PIL import Image
    arr = t[0].detach().cpu().permute(1, 2, 0).clip(0, 255).to(torch.uint8).numpy()
    return Image.fromarray(arr)


class _Saver:
    """JPEG writes of out.py:260-312.  The uint8 conversion runs on the GPU (clip + truncation, as `to_pillow_fn` does on the
    host), the bytes are copied to the host in the caller's thread, the JPEG encode + file write go to ``pool`` when one is given
    (``main``'s loop: the encoder releases the GIL, so the next pair's kernels are enqueued meanwhile)."""

    def __init__(self, pool=None):
        self.pool, self.futures = pool, []

    def _write(self, arr, path):
        from PIL import Image
        Image.fromarray(arr).save(path)

    def image(self, t, path):
        arr = t[0].detach().clip(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
        self.array(arr, path)

    def array(self, arr, path):
        if self.pool is None:
            self._write(arr, path)
        else:
            self.futures.append(self.pool.submit(self._write, arr, path))

    def wait(self):
        for f in self.futures:
            f.result()
        self.futures = []


@torch.no_grad()
def load_inpainter(name):
    """out.py:341-346: `core.inference.mix_methods.utils.<name>.inpainter`.  The reference's inpainters (TransRef, diffusion)
    need fetched weights and third-party CUDA ops and are out of scope; a module of that name dropped into
    `stitch_amd/mix_methods/utils/` is picked up, otherwise the pass-through stand-in is used."""
    import importlib
    try:
        return importlib.import_module(f"stitch_amd.mix_methods.utils.{name}").inpainter
    except ImportError:
        print(f"[out.py] inpainter {name!r} is not available here (out of scope): using the pass-through stand-in -- holes keep "
              f"what mix_fn fills from image 1, the thin border it leaves to the inpainter stays as is")
        return importlib.import_module("stitch_amd.mix_methods.utils.passthrough_inpainter").inpainter


def inference_one_data(cfg, data_dict, save_root_path, warp_model, composition_model=None, inpainter=None, forward=None, saver=None):
    """out.py:158-312: forward (`test_out`), TPS post-pipeline with the configured `mix_fn`, saves, composition.  The neural
    inpainter the reference calls inside `mix_fn` is out of scope (pass-through stand-in unless the caller supplies one): in
    `warp2.jpg`, `mask2.jpg`, `ave_fusion.jpg` and the composition inputs the holes hold what `mix_fn` fills from image 1, the
    thin border region it hands to the inpainter is not synthesised.

    ``forward``: a callable returning the `test_out` dict of this pair whose network part is already in flight (``main``
    launches pair i + 1's hipGraph before it finishes pair i); default = load + ``warp_model(..., type="test_out")`` here.
    ``saver``: a ``_Saver`` (JPEG encodes on a thread pool); default = write synchronously."""
    saver = saver or _Saver()
    path = data_dict["DATA_PATH"]
    name = os.path.basename(os.path.normpath(path))
    result_path = os.path.join(save_root_path, name) + "/"
    os.makedirs(result_path, exist_ok=True)
    if forward is not None:
        out = forward()
    else:
        image1, image2 = uploadSingleData(decodeSingleData(path if path.endswith("/") else path + "/", data_dict["IMG1"], data_dict["IMG2"]),
                                          resize_to_512=cfg.resize_to_512)
        if getattr(cfg, "swap_image", False):
            image1, image2 = image2, image1
        out = warp_model(image1, image2, type="test_out", pad_mode=cfg.pad_mode)
    if inpainter is None:
        inpainter = load_inpainter(getattr(cfg.TPS_PIPELINE_CONFIG, "inpainter", "") or "passthrough_inpainter")
    # ---- TPS post-pipeline incl. the mix_fn plug-in (out.py:218-258, core/inference/tps_pipline.py:20-205) on the GPU
    import stitch_amd
    tpc = cfg.TPS_PIPELINE_CONFIG
    fb = getattr(cfg, "use_fb_consistency_mask", False)
    valid = out["origin_occlusion_mask"] if (fb and tpc.use_valid_on_flow) else None                      # out.py:219-222
    border_points_mask = None
    if fb and tpc.use_border_points_mask:                                                                 # out.py:224-232
        border_points_mask = out["occlusion_mask"] if tpc.use_occ_filter else (out["H_warp_mask"].mean(dim=1, keepdim=True) > 0.5).float()
    inputs = dict(output1=out["output1"], mask1=out["mask1"], H_warp=out["H_warp"], H_warp_mask=out["H_warp_mask"],
                  final_warp=out["final_warp"], mask2=out["mask2"], residual_flow=out["residual_flow"], valid=valid,
                  occlusion_mask=out["occlusion_mask"], border_points_mask=border_points_mask)
    limit = dict(width_min=out["width_min"], height_min=out["height_min"], out_height=out["out_height"], out_width=out["out_width"])
    import importlib
    mix_fn = importlib.import_module(f"stitch_amd.mix_methods.{tpc.mix_method}").mix_fn                     # out.py:235
    inpaint_fn = lambda **kw: mix_fn(**kw, inpainter=inpainter, use_composition=tpc.use_composition_when_inpaint,   # noqa: E731
                                     is_plot=tpc.is_plot, resize_to_area_limit_before_inpaint=tpc.resize_to_area_limit_before_inpaint)
    new = stitch_amd.tps_pipeline.tps_H_warp(inputs, limit, tpc, inpaint_fn=inpaint_fn)
    out = dict(out, forward_output2=out["output2"], forward_mask2=out["mask2"], forward_blend_image=out["blend_image"],
               new_blend_image=new["new_blend_image"], tps_output=new["tps_output"], output2=new["output2"],
               mask2=new["mask2"] if new["mask2"].shape[1] == 3 else new["mask2"].expand(-1, 3, -1, -1))   # 1 channel normally; the
    #                      mix_fn branch "inpaint result all zero: not used" returns a 3-channel mask (reference: passed through as is)
    saver.image(out["H_warp"], result_path + "H_warp.jpg")
    saver.image(out["final_warp"], result_path + "flow_warp.jpg")
    saver.image(out["output1"], result_path + "warp1.jpg")
    saver.image(out["output2"], result_path + "warp2.jpg")                                                # out.py:265-272
    for key in ("mask1", "mask2"):
        saver.array((out[key] > 0.5)[0, 0].to(torch.uint8).mul(255).cpu().numpy(), result_path + key + ".jpg")
    saver.image(out["new_blend_image"].float(), result_path + "ave_fusion.jpg")
    if composition_model is not None:
        # out.py:277-312: learned seam masks + composed image from the UDIS2 composition network
        mask1, mask2 = (out["mask1"] > 0.5).float(), (out["mask2"] > 0.5).float()
        comp = stitch_amd.composition.compose(composition_model, out["output1"], out["output2"], mask1, mask2)
        saver.image((comp["stitched_image"] + 1) * 127.5, result_path + "composition.jpg")
        for key in ("learned_mask1", "learned_mask2"):
            saver.image(comp[key] * 255, result_path + key + ".jpg")
        out = dict(out, **comp)
    return out, result_path


def run_pairs(cfg, todo, save_root, model, composition_model=None, inpainter=None, on_done=None, depth=2):
    """The inference loop of out.py:351-357 as a software pipeline, `depth` pairs in flight.  While pair i is finished on the host
    (canvas bounds read back, canvas kernels, TPS post-pipeline with its control-point round trips, composition, device->host copies
    of the images), the network parts of pairs i + 1 .. i + depth - 1 -- both nets at 512x512, ~1 000 launches each, replayed from a
    hipGraph (`GraphedTestOut.launch`) -- are already running on their own HIP streams, the JPEGs of the pairs after them are being
    decoded, and pair i - 1's JPEGs are being encoded, both on worker threads.  `depth` graph objects alternate so that a pair's static
    buffers (`residual_flow`, ...) are not overwritten before its post-pipeline has consumed them.  Same kernels and the same files
    as calling `inference_one_data` pair by pair (tests/test_harness_gpu.py).  Measured on 48 synthetic 512x512 pairs, 10 JPEGs written per
    pair (tools/bench_out_harness.py, profiles/r4_out_harness.json): 36.9 pairs/s pair by pair, 60.0 at depth 2 (default), 51 at depth 3 / 4
    (a third network graph in flight only delays the canvas / TPS / composition kernels of the pair the host is waiting for)."""
    from concurrent.futures import ThreadPoolExecutor
    if not todo:
        return []
    depth = max(1, int(depth))
    graphs = [model.graphed_test_out() for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    done = []
    with ThreadPoolExecutor(max_workers=2) as dec_pool, ThreadPoolExecutor(max_workers=4) as enc_pool:
        def decode(dd):
            p = dd["DATA_PATH"]
            return decodeSingleData(p if p.endswith("/") else p + "/", dd["IMG1"], dd["IMG2"])

        decoded = [dec_pool.submit(decode, dd) for dd in todo[:depth + 1]]

        def launch(j):
            k = j % depth
            arrays = decoded[j].result()
            decoded[j] = None                  # the Future keeps both uint8 arrays alive: release it (the reference's loop holds one pair)
            if j + depth + 1 < len(todo):
                decoded.append(dec_pool.submit(decode, todo[j + depth + 1]))
            with torch.cuda.stream(streams[k]):
                image1, image2 = uploadSingleData(arrays, resize_to_512=cfg.resize_to_512)
                if getattr(cfg, "swap_image", False):
                    image1, image2 = image2, image1
                return graphs[k], graphs[k].launch(image1, image2), streams[k]

        inflight = [launch(j) for j in range(min(depth - 1, len(todo)))]
        saver = _Saver(enc_pool)
        for j, dd in enumerate(todo):
            if j + depth - 1 < len(todo):
                inflight.append(launch(j + depth - 1))
            g, handle, st = inflight.pop(0)
            with torch.cuda.stream(st):
                out, rp = inference_one_data(cfg, dd, save_root, model, composition_model, inpainter,
                                             forward=lambda: g.finish(handle), saver=saver)
            # no host wait here: the graph of this slot is launched again on the SAME stream (pair j + depth), i.e. after everything
            # that reads this pair's static buffers
            print("saved", rp)
            done.append(rp)
            if on_done is not None:
                on_done(j, out)
        saver.wait()
    return done


def shard_of_this_rank(data, rank, world):
    """Pairs are independent: under ``torchrun --nproc-per-node N out.py ...`` rank r takes pairs r, r + N, ... of the list
    (`stitch_amd.dist.shard_indices`; replaces the reference's single-process nn.DataParallel, out.py:80) and every rank
    writes its own result directories; no collective is needed."""
    from stitch_amd import dist as sdist
    return [data[i] for i in sdist.shard_indices(len(data), rank, world)]


def main(argv=None):
    import stitch_amd
    from stitch_amd import dist as sdist
    cfg = get_config(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if getattr(cfg, "dry_run", False):
        rank = int(os.environ.get("RANK", "0"))
        mine = shard_of_this_rank(get_data_dict_list(cfg.data_root_path, cfg.txt_file), rank, world)
        print("DRY_RUN rank", rank, "of", world, "pairs", [os.path.basename(os.path.normpath(d["DATA_PATH"])) for d in mine], flush=True)
        return None
    if world > 1:
        rank, world, local = sdist.init()              # one process per GPU: binds this process to cuda:LOCAL_RANK
    else:
        rank, local = 0, cfg.gpu
        torch.cuda.set_device(cfg.gpu)
    model = stitch_amd.build_model(cfg)
    if cfg.restore_ckpt:
        model.load_state_dict(torch.load(cfg.restore_ckpt, map_location="cpu", weights_only=True), strict=True)
    else:
        print("[out.py] no --restore_ckpt given: running with random-init weights (plumbing only)")
    model = model.cuda().eval()
    composition_model = None
    if getattr(cfg, "use_composition", False):
        import stitch_amd
        path = getattr(cfg, "composition_model_path", "")
        if path and os.path.exists(path):
            composition_model, _ = stitch_amd.composition.load_com_model(path)                 # out.py:95-103
        else:
            print(f"[out.py] composition checkpoint {path!r} not found: random-init composition network (plumbing only)")
            composition_model = stitch_amd.composition.Network().cuda().eval()
    inpainter = load_inpainter(getattr(cfg.TPS_PIPELINE_CONFIG, "inpainter", "") or "passthrough_inpainter")   # out.py:339-346
    model_name = cfg.restore_ckpt.split("/")[-2] if cfg.restore_ckpt.count("/") >= 1 else "random"
    tag = "512" if cfg.resize_to_512 else ""
    tps = cfg.TPS_PIPELINE_CONFIG
    suffix = f"{tag}_{model_name}_{tps.get_pt_methods[0]}_{tps.mix_method}_g{tps.grid_h}"
    save_root = os.path.abspath(os.path.join(cfg.data_root_path, f"../{cfg.result_dir}/")) + f"/ours_{suffix}/"
    os.makedirs(save_root, exist_ok=True)
    with open(save_root + "config.txt", "w") as f:
        f.write(repr(dict(cfg)))
    todo = []
    for dd in shard_of_this_rank(get_data_dict_list(cfg.data_root_path, cfg.txt_file), rank, world):
        if cfg.skip_if_avg_fusion_exists and os.path.exists(os.path.join(save_root, os.path.basename(os.path.normpath(dd["DATA_PATH"])), "ave_fusion.jpg")):
            print("[WARNING] Skip, Due to exist", dd["DATA_PATH"])
            continue
        todo.append(dd)
    run_pairs(cfg, todo, save_root, model, composition_model, inpainter)
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    return save_root


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import stitch_amd                                   # noqa: F401
        from stitch_amd import dist as _sdist
        with _sdist.rank_guard("out.py"):
            main()
    else:
        main()
