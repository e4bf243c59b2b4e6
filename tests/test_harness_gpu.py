"""The product harnesses as software pipelines: `stitch_amd.evaluate.validate_with_model`
(reference evaluate.py:23-65) with hipGraph replay + k streams + decode-ahead, and the `out.py` loop (reference out.py:158-216,
351-357) with pair i + 1's network graph in flight while pair i is finished.  Both must return the SAME bits / files as the plain
one-pair-at-a-time loops: same kernels, another schedule."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(seeded_sd):
    import stitch_amd
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    m = stitch_amd.build_model(cfg)
    m.load_state_dict(seeded_sd, strict=True)
    return m.cuda().eval()


def _write_split(root, n, size=(512, 512), grey=()):
    from PIL import Image
    from stitch_amd.data import structured_pair
    for d in ("input1", "input2"):
        os.makedirs(os.path.join(root, "testing", d), exist_ok=True)
    for i in range(n):
        a, b = structured_pair(size[0], size[1], seed=300 + i, shift=(2 * i - 5, 7 - 3 * i))
        for d, t in (("input1", a), ("input2", b)):
            arr = t[0].permute(1, 2, 0).numpy().astype(np.uint8)
            im = Image.fromarray(arr)
            if i in grey:
                im = im.convert("L")              # core/datasets.py tiles a grey image to 3 channels
            im.save(os.path.join(root, "testing", d, f"{i:06d}.jpg"), quality=95)


def test_load_rgb8_is_the_host_conversion():
    """st_load_rgb8 == torch.from_numpy(img).permute(2, 0, 1).float() (core/datasets.py:383-386), bit for bit."""
    import stitch_amd
    g = torch.Generator().manual_seed(5)
    for shape in ((1, 512, 512, 3), (3, 30, 44, 3), (2, 7, 4, 3)):
        u8 = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        got = stitch_amd.ops.load_rgb8(u8.cuda())
        assert torch.equal(got.cpu(), u8.permute(0, 3, 1, 2).float())
    with pytest.raises(stitch_amd.ops.StitchErrorBase):
        stitch_amd.ops.load_rgb8(torch.zeros((1, 5, 5, 3), dtype=torch.uint8).cuda())      # H*W not a multiple of 4: rejected, not mis-read


def test_graph_replays_are_stateless(model):
    """Every replay of a captured forward must return the eager forward's bits, whatever ran before it.  Round 4 found the
    range map's u64 accumulator cleared by `hipMemsetAsync`: captured as a memset node it did NOT clear on replays, so from the
    second replay on the occlusion mask (hence `final_warp_output`) was the sum of two splats.  The accumulator is now zeroed by
    a kernel; this test replays three times, with other pairs and an eager call in between."""
    from stitch_amd.data import structured_pair
    pairs = [tuple(t.cuda() for t in structured_pair(512, 512, seed=40 + i)) for i in range(2)]
    eager = [model(a, b, type="test_eval") for a, b in pairs]
    g = model.graphed("test_eval")
    for rep in range(3):
        for (a, b), ref in zip(pairs, eager):
            o = g(a, b)
            for k, v in ref.items():
                got, want = (o[k][0], v[0]) if isinstance(v, list) else (o[k], v)
                assert torch.equal(got, want), (rep, k, (got - want).abs().max().item())
        model.predict_homo(*pairs[rep % 2])                      # an eager call between replays changes nothing
    gt = model.graphed_test_out()
    a, b = (t.cuda() for t in structured_pair(304, 400, seed=77, shift=(3, -4)))
    ref = model(a, b, type="test_out")
    for rep in range(3):
        o = gt(a, b)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert torch.equal(o[k], v), (rep, k)
            else:
                assert o[k] == v, (rep, k)


def test_pipelined_eval_harness_equals_the_plain_loop(tmp_path, model):
    """k pairs in flight from hipGraphs + decode-ahead + metric kept on the device == eager loop with a .cpu() per pair
    (`torch.equal` on the per-pair (psnr, ssim) table), incl. a grey JPEG, more pairs than slots, and a second call that reuses
    the captured graphs."""
    from stitch_amd import evaluate as ev
    _write_split(str(tmp_path), 8, grey=(5,))
    ds = ev.UDISDataset(str(tmp_path) + "/", phase="testing")
    assert len(ds) == 8
    res_p, tab_p = ev.validate_with_model(model, ds, streams=3)
    res_e, tab_e = ev.validate_with_model(model, ds, pipelined=False)
    assert tab_p.shape == (8, 2) and torch.isfinite(tab_p).all()
    assert torch.equal(tab_p, tab_e), (tab_p - tab_e).abs().max()
    assert str(res_p) == str(res_e)                 # (NaN entries: fewer than 663 pairs leave the mid / hard slices empty)
    pipe = model._eval_pipeline
    n_graphs = pipe.n_captured()
    assert n_graphs == 3 and len(pipe._slots) == 1
    _, tab_again = ev.validate_with_model(model, ds, streams=3)            # replays the same graphs
    assert torch.equal(tab_again, tab_p) and model._eval_pipeline.n_captured() == n_graphs and model._eval_pipeline is pipe
    # batches (evaluate.py:34 uses 12): a full batch of 3 and a ragged tail of 2 are two graph shapes; rows keep dataset order
    _, tab_b = ev.validate_with_model(model, ds, batch_size=3, streams=2)
    _, tab_be = ev.validate_with_model(model, ds, batch_size=3, pipelined=False)
    assert torch.equal(tab_b, tab_be)


def test_graphs_are_recaptured_when_the_weights_change_identity(tmp_path, model, seeded_sd):
    """a captured graph reads the PACKED copies of the weights it was recorded with.  `model.flow_backbone.load_state_dict`
    (a sub-module's own load: the adapter's override never runs), `.float()` / `.to()` (an `_apply`, even a no-op) and in-place parameter
    writes all leave those copies dead or stale.  Every graph holder compares `FlowHomoAdpater.weights_generation()` before replaying and
    re-captures on mismatch: after each change the pipelined harness, `GraphedForward` and `GraphedTestOut` must equal the eager path."""
    from stitch_amd import evaluate as ev
    from stitch_amd.data import structured_pair
    _write_split(str(tmp_path), 3)
    ds = ev.UDISDataset(str(tmp_path) + "/", phase="testing")
    _, tab0 = ev.validate_with_model(model, ds, streams=2)
    g, gt = model.graphed("test_eval"), model.graphed_test_out()
    a, b = (t.cuda() for t in structured_pair(512, 512, seed=41))
    c, d = (t.cuda() for t in structured_pair(304, 400, seed=77, shift=(3, -4)))
    g(a, b), gt(c, d)
    prefix = "flow_backbone."
    other = {k[len(prefix):]: (v * 0.97 if v.dtype.is_floating_point and v.ndim >= 2 else v) for k, v in seeded_sd.items() if k.startswith(prefix)}
    try:
        gen0 = model.weights_generation()
        model.flow_backbone.load_state_dict(other, strict=True)                    # not through FlowHomoAdpater.load_state_dict
        assert model.weights_generation() != gen0
        _, tab_p = ev.validate_with_model(model, ds, streams=2)
        _, tab_e = ev.validate_with_model(model, ds, pipelined=False)
        assert torch.equal(tab_p, tab_e) and not torch.equal(tab_p, tab0)
        ref = model(a, b, type="test_eval")
        o = g(a, b)
        assert torch.equal(o["final_warp_output"], ref["final_warp_output"]) and torch.equal(o["flow_predictions"][0], ref["flow_predictions"][0])
        ref_t = model(c, d, type="test_out")
        o_t = gt(c, d)
        assert torch.equal(o_t["blend_image"], ref_t["blend_image"]) and torch.equal(o_t["residual_flow"], ref_t["residual_flow"])
        # a no-op _apply drops the packed copies as well
        model.float()
        o = g(a, b)
        assert torch.equal(o["final_warp_output"], ref["final_warp_output"])
        # an in-place write to one parameter (no hook fires): seen by the harness's deep check at the start of a run
        with torch.no_grad():
            w = next(p for n, p in model.homo_backbone.named_parameters() if p.ndim == 4)
            w.mul_(1.01)
        # (no manual _invalidate(): the deep check drops the backbones' packed copies itself -- ADVICE r5)
        _, tab_p2 = ev.validate_with_model(model, ds, streams=2)
        assert not torch.equal(tab_p2, tab_p), "the harness still computes with the weights packed before the in-place write"
        model.homo_backbone._invalidate(); model.flow_backbone._invalidate()       # the reference: eager run on freshly packed weights
        _, tab_e2 = ev.validate_with_model(model, ds, pipelined=False)
        assert torch.equal(tab_p2, tab_e2)
        o = g(a, b)                                                                 # the shallow holders re-capture too (the generation moved)
        ref2 = model(a, b, type="test_eval")
        assert torch.equal(o["final_warp_output"], ref2["final_warp_output"]) and not torch.equal(ref2["final_warp_output"], ref["final_warp_output"])
    finally:
        model.load_state_dict(seeded_sd, strict=True)
    _, tab_back = ev.validate_with_model(model, ds, streams=2)
    assert torch.equal(tab_back, tab0)


def test_eval_pipeline_keeps_a_bounded_number_of_shapes(tmp_path, model):
    """the graph cache is an LRU over input shapes (`EvalPipeline.MAX_SHAPES`), not an ever-growing dict.  The nets run
    at 512x512 only (`test_eval`), so the shapes here are batch sizes: batches of 1..5 with their ragged tails are 5 distinct shapes."""
    from stitch_amd import evaluate as ev
    _write_split(str(tmp_path), 5)
    ds = ev.UDISDataset(str(tmp_path) + "/", phase="testing")
    model._eval_pipeline = None
    tabs = {}
    for bs in (1, 2, 3, 4, 5):
        _, tabs[bs] = ev.validate_with_model(model, ds, batch_size=bs, streams=2)
        assert len(model._eval_pipeline._slots) <= ev.EvalPipeline.MAX_SHAPES
    pipe = model._eval_pipeline
    # batch sizes 2 and 4 leave a tail of one pair (shape 1 stays recently used); the least recently used shape, 3, went
    assert len(pipe._slots) == ev.EvalPipeline.MAX_SHAPES and {k[0][0] for k in pipe._slots} == {1, 2, 4, 5}
    _, again = ev.validate_with_model(model, ds, batch_size=3, streams=2)           # evicted shape, captured again: same bits
    assert torch.equal(again, tabs[3]) and model._eval_pipeline is pipe and len(pipe._slots) == ev.EvalPipeline.MAX_SHAPES
    model._eval_pipeline = None


def test_pipelined_eval_mixed_shapes_and_generic_dataset(tmp_path, model):
    """a batch whose pairs differ in shape runs pair by pair (as the plain loop does); a dataset without `load_u8` (the
    reference's `__getitem__` protocol: float [3,H,W] tensors) is uploaded as floats."""
    from stitch_amd import evaluate as ev
    _write_split(str(tmp_path / "a"), 2)
    ds = ev.UDISDataset(str(tmp_path / "a") + "/", phase="testing")

    class Plain:                                    # the reference's dataset protocol only
        def __len__(self):
            return len(ds)

        def __getitem__(self, i):
            return ds[i]

    _, t_u8 = ev.validate_with_model(model, ds, streams=2)
    _, t_f32 = ev.validate_with_model(model, Plain(), streams=2)
    assert torch.equal(t_u8, t_f32)


def _load_out_module():
    spec_ = importlib.util.spec_from_file_location("stitch_out_harness_p", os.path.join(ROOT, "out.py"))
    outmod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(outmod)
    return outmod


def test_out_loop_pipelined_writes_the_same_files(tmp_path, model):
    """`run_pairs` (graph of pair i + 1 in flight, JPEG decode / encode on worker threads) writes byte-identical files to
    calling `inference_one_data` pair by pair (eager `test_out`), on 3 pairs of two different sizes."""
    from PIL import Image
    import stitch_amd
    from stitch_amd.data import structured_pair
    outmod = _load_out_module()
    root = tmp_path / "demo"
    names = []
    for i, size in enumerate(((256, 256), (304, 400), (256, 256))):
        d = root / f"p{i}"
        d.mkdir(parents=True)
        a, b = structured_pair(size[0], size[1], seed=500 + i, shift=(3 - i, 2 * i - 4))
        Image.fromarray(a[0].permute(1, 2, 0).numpy().astype(np.uint8)).save(str(d / "input1.jpg"), quality=97)
        Image.fromarray(b[0].permute(1, 2, 0).numpy().astype(np.uint8)).save(str(d / "input2.jpg"), quality=97)
        names.append(f"p{i}/")
    (root / "demo.txt").write_text("\n".join(names) + "\n")
    cfg = outmod.get_config(["--data_root_path", str(root) + "/"])
    todo = outmod.get_data_dict_list(cfg.data_root_path, cfg.txt_file)
    comp = stitch_amd.composition.Network().cuda().eval()
    inp = outmod.load_inpainter("passthrough_inpainter")
    dir_a, dir_b = str(tmp_path / "piped") + "/", str(tmp_path / "plain") + "/"
    os.makedirs(dir_a), os.makedirs(dir_b)
    done = outmod.run_pairs(cfg, todo, dir_a, model, comp, inp)
    assert len(done) == 3
    for dd in todo:
        outmod.inference_one_data(cfg, dd, dir_b, model, comp, inp)
    for i in range(3):
        fa, fb = sorted(os.listdir(dir_a + f"p{i}")), sorted(os.listdir(dir_b + f"p{i}"))
        assert fa == fb and len(fa) == 10
        for f in fa:
            assert open(dir_a + f"p{i}/" + f, "rb").read() == open(dir_b + f"p{i}/" + f, "rb").read(), (i, f)
