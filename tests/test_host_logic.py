"""Host-side mirror of the reference interface: validation, error behaviour, knot slicing, masks,
state-dict compatibility -- everything that does not need a GPU."""
import numpy as np
import pytest
import torch

from curl_amd import colors, curves, model, ops, shard, transpose


def test_cpu_tensors_are_refused_not_silently_computed():
    x = torch.rand(1, 3, 4, 4)
    for fn in (ops.rgb2lab, ops.lab2rgb, ops.rgb2hsv, ops.hsv2rgb):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        curves.apply_curve(x, torch.ones(1, 16), torch.zeros(1), 0, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.CURLLayer()(x, None, torch.zeros(1, 48), torch.zeros(1, 48), torch.zeros(1, 64))


def test_shape_and_dtype_validation(monkeypatch):
    with pytest.raises(TypeError):
        ops._image(np.zeros((1, 3, 2, 2)))
    monkeypatch.setattr(ops, "_need_device", lambda t, name: None)
    with pytest.raises(ValueError, match=r"\[B,3,H,W\]"):
        ops._image(torch.zeros(1, 4, 2, 2))
    with pytest.raises(ValueError):
        ops._image(torch.zeros(3, 2, 2))
    with pytest.raises(TypeError, match="float32"):
        ops._image(torch.zeros(1, 3, 2, 2, dtype=torch.float64))
    with pytest.raises(ValueError, match="empty"):
        ops._image(torch.zeros(0, 3, 2, 2))
    x = torch.zeros(2, 3, 4, 6)[:, :, :, ::2]  # non-contiguous view is accepted and made contiguous
    assert ops._image(x).is_contiguous()


def test_knot_validation_helpers(monkeypatch):
    monkeypatch.setattr(ops, "_need_device", lambda t, name: None)
    k, K = ops._knots(torch.zeros(2, 48), "R", 3, 2)
    assert K == 16 and k.is_contiguous()
    k, K = ops._knots(torch.zeros(2, 100)[:, :64], "H", 4, 2)  # the slice H[:, :64] of a wider head
    assert K == 16 and k.is_contiguous()
    # torch.chunk's uneven split (curves.py:53,105,152): 50 -> 17, 17, 16, packed as K | K_last << 16 (CURL_K_UNEVEN)
    _, K = ops._knots(torch.zeros(2, 50), "R", 3, 2)
    assert K == 17 | (16 << 16)
    _, K = ops._knots(torch.zeros(2, 62), "H", 4, 2)
    assert K == 16 | (14 << 16)
    for n, nc in ((3, 3), (13, 4), (4, 3), (9, 4)):  # one-knot curves; counts torch.chunk splits into fewer chunks than curves
        with pytest.raises(ValueError, match="knots per curve|torch.chunk"):
            ops._knots(torch.zeros(2, n), "R", nc, 2)
    for n, nc in ((47, 3), (62, 4), (50, 3), (8, 3), (160, 4)):  # the split IS torch.chunk's
        _, K = ops._knots(torch.zeros(2, n), "R", nc, 2)
        sizes = [c.shape[1] for c in torch.chunk(torch.zeros(2, n), nc, dim=1)]
        assert sizes == [K & 0xffff] * (nc - 1) + [(K >> 16) or (K & 0xffff)]
    with pytest.raises(ValueError):
        ops._knots(torch.zeros(3, 48), "R", 3, 2)


def test_mask_handling(monkeypatch):
    monkeypatch.setattr(ops, "_need_device", lambda t, name: None)
    img = torch.zeros(2, 3, 4, 5)
    assert ops._mask(None, img) == (None, 0)
    m, kind = ops._mask(torch.ones(2, 1, 4, 5, dtype=torch.bool), img)
    assert kind == 1 and m.dtype == torch.uint8
    m, kind = ops._mask(torch.ones(1, 1, 4, 5), img)  # broadcast over the batch like torch would
    assert kind == 2 and m.shape == (2, 1, 4, 5) and m.is_contiguous()
    m, kind = ops._mask(torch.ones(2, 1, 4, 5, dtype=torch.float64), img)
    assert kind == 2 and m.dtype == torch.float32
    m, kind = ops._mask(torch.ones(2, 4, 5), img)
    assert m.shape == (2, 1, 4, 5)
    with pytest.raises(ValueError):
        ops._mask(torch.ones(2, 3, 4, 5), img)
    with pytest.raises(TypeError):
        ops._mask(torch.ones(2, 1, 4, 5, dtype=torch.int32), img)


def test_state_dict_keys_match_reference():
    """Checkpoints of the reference carry the colour constants under these keys."""
    layer = model.CURLLayer()
    keys = set(layer.state_dict())
    want = {"rgb2lab.rgb_to_xyz", "rgb2lab.fxfyfz_to_lab", "rgb2lab.xyz_to_rgb_mult", "rgb2lab.lab_to_fxfyfz_offset",
            "lab2rgb.xyz_to_rgb", "lab2rgb.lab_to_fxfyfz", "lab2rgb.xyz_to_rgb_mult", "lab2rgb.lab_to_fxfyfz_offset",
            "rgb2hsv.comparison_zero"}
    assert want == keys
    assert layer.state_dict()["rgb2lab.rgb_to_xyz"].shape == (1, 1, 3, 3)
    assert not any(p.requires_grad for p in layer.parameters())


def test_colour_constants_match_reference_modules(reference_modules):
    ref = reference_modules["colors"]
    for mine, theirs in ((colors.RGB2LAB(), ref.RGB2LAB()), (colors.LAB2RGB(), ref.LAB2RGB()),
                         (colors.RGB2HSV(), ref.RGB2HSV())):
        a, b = mine.state_dict(), theirs.state_dict()
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_gcurlnet_structure_and_split():
    net = model.GCURLNet(backbone=model.CurveEncoder(num_outputs=160, width=0.25, num_features=64)).eval()
    assert (net.curve_break_1, net.curve_break_2) == (48, 96)  # model.py:186-187
    knots = net.predict_knots(torch.rand(2, 3, 64, 64))
    assert knots.shape == (2, 160)
    # an injected backbone with a mismatched Linear head gets the 160-wide head of model.py:190-192
    bb = model.CurveEncoder(num_outputs=10, width=0.25, num_features=64)
    net2 = model.GCURLNet(backbone=bb).eval()
    assert net2.predict_knots(torch.rand(1, 3, 32, 32)).shape == (1, 160)
    net3 = model.GCURLNet(backbone=model.CurveEncoder(160, 0.25, 64), encoder_size=32).eval()
    assert net3.predict_knots(torch.rand(1, 3, 100, 75)).shape == (1, 160)


def test_transpose_matches_golden(golden):
    g = golden("layout")
    assert np.array_equal(transpose.swapimdims_3HW_HW3(g["chw"]), g["chw_to_hwc"])
    assert np.array_equal(transpose.swapimdims_3HW_HW3(g["bchw"]), g["bchw_to_bhwc"])
    assert np.array_equal(transpose.swapimdims_HW3_3HW(g["chw_to_hwc"]), g["hwc_to_chw"])
    assert np.array_equal(transpose.swapimdims_HW3_3HW(g["bchw_to_bhwc"]), g["bhwc_to_bchw"])
    assert transpose.swapimdims_3HW_HW3(np.zeros((2, 2))) is None  # reference returns None for other ranks


def test_image_shard_is_a_partition():
    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard.image_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.image_shard(4, 2, 2)


def test_convert_state_dict_matches_reference(reference_modules):
    import importlib
    import sys
    from conftest import REFERENCE
    from curl_amd.convert_state import convert_state_dict
    sys.path.insert(0, REFERENCE)
    try:
        ref = importlib.import_module("convert_state")
    finally:
        sys.path.remove(REFERENCE)
    g = torch.Generator().manual_seed(0)
    sd = {"module.rgb2lab.rgb_to_xyz": torch.rand(3, 3, generator=g), "module.lab2rgb.lab_to_fxfyfz": torch.rand(3, 3, generator=g),
          "lab2rgb.xyz_to_rgb": torch.rand(1, 1, 3, 3, generator=g), "module.backbone.classifier.weight": torch.rand(4, 5, generator=g),
          "rgb2hsv.comparison_zero": torch.tensor(0.0)}
    a, b = convert_state_dict(sd), ref.convert_state_dict(sd)
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_reference_style_checkpoint_loads_into_curllayer():
    """A DDP-saved state dict with old 2-D colour matrices loads into the mirror layer key for key."""
    from curl_amd.convert_state import convert_state_dict
    layer = model.CURLLayer()
    old_2d = ("rgb2lab.rgb_to_xyz", "rgb2lab.fxfyfz_to_lab", "lab2rgb.xyz_to_rgb", "lab2rgb.lab_to_fxfyfz")
    sd = {"module." + k: (v[0, 0].t().clone() if k in old_2d else v.clone()) for k, v in layer.state_dict().items()}
    layer2 = model.CURLLayer()
    layer2.load_state_dict(convert_state_dict(sd))
    for k, v in layer.state_dict().items():
        assert torch.equal(v, layer2.state_dict()[k])


def test_folder_dataset_mirror(tmp_path):
    """curl_amd.data: the reference's folder layout (data.py:43-80), one transform for input / output / mask, item keys
    and dtypes of data.py:176-207, centre crop in evaluation mode, zero padding of too-small images."""
    import importlib.util
    import os
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("curl_amd_data_only", os.path.join(root, "curl_amd", "data.py"))
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    rng = np.random.default_rng(0)
    for d in ("curl_input", "curl_output", "masks"):
        os.makedirs(tmp_path / d)
    for i, (h, w) in enumerate([(80, 96), (40, 50), (300, 280)]):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(rgb).save(tmp_path / "curl_input" / f"{i}.png")
        Image.fromarray(rgb).save(tmp_path / "curl_output" / f"{i}.png")          # output == input: must stay equal
        m = np.zeros((h, w), np.uint8)
        m[h // 4:, w // 3:] = 255
        Image.fromarray(m, "L").save(tmp_path / "masks" / f"{i}.png")
    (tmp_path / "images_train.txt").write_text("0\n2\n")
    dd = data.get_data_dict(os.path.join(str(tmp_path), ""))
    assert sorted(dd) == [0, 1, 2] and dd[1]["mask"].endswith("masks/1.png")
    ids = data.get_data_ids(str(tmp_path / "images_train.txt"))
    assert ids == [0, 2]
    sub = data.filter_data_dict(dd, ids)
    assert sorted(sub) == [0, 1] and sub[1] is dd[2]
    train = data.Dataset(sub, normaliser=1, is_train=True, crop_h=64, crop_w=64, seed=3)
    for k in range(4):
        item = train[k % 2]
        assert item["input_img"].shape == (3, 64, 64) and item["input_img"].dtype == torch.float32
        assert item["mask"].shape == (1, 64, 64) and item["mask"].dtype == torch.bool
        assert torch.equal(item["input_img"], item["output_img"])                    # same crop / flips / rotation
        assert 0.0 <= float(item["input_img"].min()) and float(item["input_img"].max()) <= 1.0
        assert item["name"] in ("0.png", "2.png")
    ev = data.Dataset(dd, normaliser=1, is_train=False, crop_h=64, crop_w=64)
    a, b = ev[0], ev[0]
    assert torch.equal(a["input_img"], b["input_img"])                               # deterministic centre crop
    full = torch.from_numpy(np.array(Image.open(tmp_path / "curl_input" / "0.png"))).permute(2, 0, 1).float() / 255
    assert torch.equal(a["input_img"], full[:, 8:72, 16:80])
    small = ev[1]                                                                    # 40x50 image: zero padded
    assert small["input_img"].shape == (3, 64, 64) and float(small["input_img"][:, 0, 0].abs().max()) == 0.0
    # the rotation kernel: +-90 degrees on a square is an exact quarter turn
    x = torch.arange(3 * 8 * 8, dtype=torch.float32).reshape(3, 8, 8)
    r = data._rotate_nearest(x, 90.0)
    assert torch.equal(r, torch.rot90(x, 1, (1, 2))) or torch.equal(r, torch.rot90(x, -1, (1, 2)))


def test_bench_gpus_n_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks as a child process (torch.distributed.run,
    127.0.0.1 rendezvous) before touching the GPU, and refuses a WORLD_SIZE that contradicts --gpus."""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the child's return code is ours
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher with the wrong world size: loud exit, not a silent 1-rank run
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(bench.torch.cuda, "set_device", lambda d: None)
    import torch.distributed as dist
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: None)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE 2" in str(e.value.code)


def test_ops_refuse_tensors_on_different_devices():
    """A knot / mask / out tensor on another GPU than the image must raise, not reach the kernel."""
    from unittest.mock import MagicMock

    def fake(dev):
        t = MagicMock(spec=torch.Tensor)
        t.device = torch.device(dev)
        return t
    a, b = fake("cuda:0"), fake("cuda:1")
    with pytest.raises(ValueError, match="same device"):
        ops._check_same_device([("img", a), ("L", b)])
    with pytest.raises(ValueError, match="L on cuda:1"):
        ops.curl_layer_forward(a, None, b, a, a)           # through the public entry point's decorator
    assert ops._check_same_device([("img", a), ("mask", None), ("R", fake("cuda:0"))]) == torch.device("cuda:0")
    assert ops._check_same_device([("img", torch.zeros(1)), ("R", a)]) is None   # CPU tensor: the body raises
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.rgb2lab(torch.zeros(1, 3, 4, 4))


def test_out_argument_is_validated():
    img = torch.zeros(2, 3, 8, 8)
    assert ops._check_out(torch.empty_like(img), img).shape == img.shape
    for bad in (torch.empty(2, 3, 8, 4), torch.empty(2, 3, 8, 8, dtype=torch.float64),
                torch.empty(2, 3, 8, 16)[..., ::2], torch.empty(2, 3, 8, 8, device="meta")):
        with pytest.raises(ValueError, match="out must be"):
            ops._check_out(bad, img)
    with pytest.raises(TypeError):
        ops._check_out([1, 2], img)


class _RandProbe(torch.utils.data.Dataset):
    """Reports the augmentation draws a data.Dataset makes inside DataLoader workers."""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return 4

    def __getitem__(self, i):
        return torch.tensor([self.ds._rand(), self.ds._rand()])


def test_augmentation_streams_differ_across_workers_and_epochs():
    """a seeded private generator copied into every DataLoader worker replayed ONE augmentation stream in
    all workers and all epochs; the reference's global-RNG draws differ per worker and per epoch."""
    from curl_amd import data
    ds = data.Dataset({}, is_train=True, seed=123)
    loader = torch.utils.data.DataLoader(_RandProbe(ds), batch_size=1, shuffle=False, num_workers=2)
    epochs = [torch.cat([b for b in loader]) for _ in range(2)]  # items 0,2 -> worker 0; items 1,3 -> worker 1
    e0, e1 = epochs
    assert not torch.equal(e0[0], e0[1])      # two workers, first draw each: different streams
    assert not torch.equal(e0, e1)            # next epoch: different draws
    assert len({float(v) for v in torch.cat([e0, e1]).flatten()}) == 16
    # without workers the one generator simply advances (and is reproducible from the seed)
    a = data.Dataset({}, is_train=True, seed=5)
    b = data.Dataset({}, is_train=True, seed=5)
    assert [a._rand() for _ in range(3)] == [b._rand() for _ in range(3)]


def test_bench_workload_table_and_replayed_traffic():
    """Every bench workload names its bound and carries the figures its roofline is computed from; the PMC traffic is
    replayed from the committed pass and says so."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert set(bench.WORKLOADS) >= {"layer", "layer_disk", "lab_stage", "rgb_only", "trispace", "layer_u8", "trispace_u8"}
    for name, w in bench.WORKLOADS.items():
        assert w["bound"] in ("hbm", "valu") and w["bpp"] > 0 and w["flop_px"] > 0, name
    assert bench.WORKLOADS["layer"]["bound"] == "hbm" and bench.WORKLOADS["layer"]["bpp"] == 25.0
    assert bench.WORKLOADS["trispace"]["bound"] == "valu"
    assert "layer_8bit" in bench.WORKLOADS and bench.WORKLOADS["layer_8bit"]["frag"] == "OpLayer"
    traffic, src, _util = bench.load_traffic("OpLayer")
    assert 1.15e9 < traffic < 1.3e9 and src.startswith("profiles/traffic_r") and "not this run" in src
    assert bench.load_traffic("NoSuchKernel", "no_such_workload") == (None, None, None)


def _fake_measure_record(bench, name, ms):
    """A measure() record of bench.py's shape for one workload, with awkward floats (what a real run prints)."""
    w = bench.WORKLOADS[name]
    npx = bench.workload_pixels(name, 32)
    gbps = npx * w["bpp"] / (ms * 1e-3) / 1e9
    tf = npx * w["flop_px"] / (ms * 1e-3) / 1e12
    hbm = {"achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "frac_of_measured_copy_ceiling_6585": gbps / 6585.0}
    valu = {"achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "flop_per_px": w["flop_px"]}
    roof = dict(hbm if w["bound"] == "hbm" else valu)
    roof.update({"bound": w["bound"], "traffic": 1205276956.6711411, "traffic_source": "profiles/traffic_r04.json (builder's "
                 "rocprofv3 --pmc pass, not this run)", "valu_issue_util": 0.8361234567, "achieved_wall": gbps * 0.97,
                 "frac_wall": gbps * 0.97 / 8000.0, "value_from_event_clock_Mpix_s": npx / (ms * 1e-3) / 1e6,
                 "algorithmic_bytes_per_px": w["bpp"], "px_per_launch": npx,
                 "secondary": {"bound": "valu", **valu} if w["bound"] == "hbm" else {"bound": "hbm", **hbm}})
    return {"workload": w["desc"], "value": npx / (ms * 1e-3) / 1e6 * 0.97, "ms_per_step": ms / 0.97, "device_ms_per_step": ms,
            "device_ms_per_step_min_over_ranks": ms, "roofline": roof,
            "cold_first_launch_us": 431.123456789,
            "cold_start": {"ms_per_step": 0.2634567891, "value": 1.8e5, "unit": "Mpix/s per GPU", "protocol": "x" * 120},
            "power": {"board_power_W_mean": 1399.2, "board_power_cap_W": 1400.0, "shader_clock_MHz_mean": 1931.0, "samples": 70}}


@pytest.mark.parametrize("world", [1, 8])
def test_bench_line_is_compact_strict_json_with_the_contract_keys(world):
    """round 4's 20.9 KB line could not be read back by the driver.  The stdout line is built in ONE
    function from the measure() records; with the whole workload table measured it stays under 4 KB, parses under a strict
    JSON parser (inf / nan -> null) and carries every key the contract names, `roofline` and `cpu_baseline`."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    main_res = _fake_measure_record(bench, "layer", 0.19809123456789)
    others = {n: _fake_measure_record(bench, n, 0.1888 + 0.01 * k) for k, n in enumerate(bench.WORKLOADS) if n != "layer"}
    assert len(others) >= 16
    accuracy = {"sample": "s" * 100, "max_abs_err": 2.37e-5, "frac_px_over_1e-5": 2.2e-5, "psnr_delta_db": 1.2e-7,
                "max_err_per_unit_sensitivity": 1.9e-6, "psnr_out_vs_ref_db": float("inf"), "reg_rel_err": float("nan")}
    cpu = {"value": 2.3812345678, "unit": "Mpix/s", "cores": 16, "kind": "port", "sample": "4 x 1500x1000 frames through the "
           "full chain (oracle/curl_oracle.py curl_layer, all-ones mask), median of 3, torch 2.10.0+rocm7.0 CPU, 16 threads of "
           "256 host cpus", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "host_cpus": 256,
           "legs": {"a": {"Mpix/s": 1.0, "threads": 16}}}
    meta = {"n_gpus": world, "steps": 20, "warmup": 5, "batch_per_gpu": 32, "backend": "nccl" if world > 1 else None,
            "ranks_seen": world, "gpus_visible": world}
    train = {"ms_per_step": 61.3, "images_per_s": 4170.0, "curve_layer_share_of_step": 0.012, "model": "m" * 200}
    line = bench.make_line(main_res, others, accuracy, cpu, meta, train)
    txt = bench.dump_line(line)
    assert "\n" not in txt and len(txt.encode()) <= bench.LINE_BUDGET <= 4096, len(txt)

    def refuse(tok):
        raise ValueError(tok)
    d = json.loads(txt, parse_constant=refuse)  # Infinity / NaN would raise here
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["metric"].startswith("Mpix/s through fused curve-apply") and "model" not in d["config"]
    assert d["config"]["workload"].startswith("CURLLayer.forward") and d["config"]["global_batch"] == 32 * world
    assert d["config"]["clock_settle_launches"] == bench.CLOCK_SETTLE_LAUNCHES
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and 0.7 < r["frac"] < 0.8
    assert r["traffic"] > 1e9 and r["valu_issue_util"] is not None and r["achieved_wall"] < r["achieved"]
    for n in ("lab_stage", "hsv_stage", "rgb_only", "layer_8bit"):
        assert r[f"{n}_us"] > 0 and r[f"{n}_GBps"] > 0 and 0 < r[f"{n}_frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] > 0 and c["unit"] == "Mpix/s" and "legs" not in c
    # both protocols lead the record: the figure with the settle launches and the literal one
    assert abs(d["literal_protocol_ms_per_step"] - 0.263457) < 1e-6
    assert abs(d["literal_protocol_frac"] - 48e6 * 25 / 0.263457e-3 / 1e9 / 8000) < 1e-3
    assert set(d["accuracy"]) == {"max_abs_err", "frac_px_over_1e-5", "psnr_delta_db", "max_err_per_unit_sensitivity"}
    assert "other_workloads" not in d and "end_to_end" not in d and "power" not in d
    # the strict dump maps non-finite floats to null wherever they sit
    assert json.loads(bench.dump_line({"a": float("inf"), "b": [float("nan"), 1.0]})) == {"a": None, "b": [None, 1.0]}
    # a slim run (no extras): still a valid line
    bare = bench.dump_line(bench.make_line(main_res, {}, None, None, meta))
    assert json.loads(bare)["roofline"]["frac"] == r["frac"]


def test_scaling_run_measures_the_targets_only():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert set(bench.SCALING_RUN_WORKLOADS) == {"lab_stage", "rgb_only"} and set(bench.SCALING_RUN_WORKLOADS) < set(bench.WORKLOADS)
