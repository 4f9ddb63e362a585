"""Golden vectors for the polynomial path, produced by RUNNING THE REFERENCE's classes.

/root/reference/model.py cannot be imported as a module here (it imports timm and torchvision at the top,
neither installed), so the three classes this path needs -- ChannelPolyLayer (model.py:206-333),
Deg4MobilePolyLayer (model.py:336-415) and the per-pixel methods of TriSpaceRegNet (cat_coords,
generate_residual, generate_image: model.py:487-520) -- are compiled from the reference's own source text
(ast, no edits) and executed with the reference's colors.py.  Nothing from oracle/ or curl_amd/ is used.

    python tests/golden/make_golden_poly.py      (build container only)
"""
import ast
import math
import operator as op
import os
import sys
from functools import reduce

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("CURL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import colors  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def reference_classes():
    tree = ast.parse(open(os.path.join(REF, "model.py")).read())
    ns = {"torch": torch, "nn": nn, "reduce": reduce, "op": op, "math": math, "colors": colors}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in ("ChannelPolyLayer", "Deg4MobilePolyLayer"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF, "model.py"), "exec"), ns)
    # TriSpaceRegNet's constructor needs timm; its per-pixel methods do not: lift them onto a plain module
    tri = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TriSpaceRegNet"][0]
    keep = [n for n in tri.body if isinstance(n, ast.FunctionDef) and n.name in ("cat_coords", "generate_residual", "generate_image")]
    shell = ast.ClassDef(name="TriSpacePixelPath", bases=[ast.Attribute(value=ast.Name(id="nn", ctx=ast.Load()), attr="Module", ctx=ast.Load())],
                         keywords=[], body=keep, decorator_list=[])
    ast.fix_missing_locations(shell)
    exec(compile(ast.Module(body=[shell], type_ignores=[]), os.path.join(REF, "model.py"), "exec"), ns)
    return ns


def make_tri(ns, spatial, polylayer, max_resolution=4096):
    """The attributes generate_residual reads, set exactly as TriSpaceRegNet.__init__ does (model.py:441-485)."""
    m = ns["TriSpacePixelPath"]()
    m.max_resolution = max_resolution
    m.polylayer = polylayer
    m.rgb2lab, m.lab2rgb, m.rgb2hsv, m.hsv2rgb = colors.RGB2LAB(), colors.LAB2RGB(), colors.RGB2HSV(), colors.HSV2RGB()
    m.sigmoid = nn.Sigmoid()
    if not spatial:
        m.x = torch.zeros(1, 0, 1, max_resolution)
        m.y = torch.zeros(1, 0, max_resolution, 1)
    else:
        m.x = torch.arange(0, max_resolution).reshape(1, 1, 1, max_resolution)
        m.y = torch.arange(0, max_resolution).reshape(1, 1, max_resolution, 1)
    return m


def main():
    ns = reference_classes()
    CPL, D4 = ns["ChannelPolyLayer"], ns["Deg4MobilePolyLayer"]
    g = torch.Generator().manual_seed(31)
    store = {}
    for d, v in ((4, 5), (4, 3), (3, 2)):
        store[f"powers_d{d}_v{v}"] = np.array(list(CPL.generate_powers(d, v)), dtype=np.int32)
    B, H, W = 2, 20, 28
    x5 = torch.rand(B, 5, H, W, generator=g)
    c5 = torch.randn(B, 3, 126, generator=g) * 0.5
    store["x5"], store["c5"] = x5.numpy(), c5.numpy()
    store["channel_poly_d4v5"] = CPL(degree=4, num_variables=5, num_out=3)(x5, c5).numpy()
    store["mobile_poly"] = D4()(x5, c5).numpy()
    x3 = torch.rand(B, 3, H, W, generator=g)
    c3 = torch.randn(B, 3, 35, generator=g) * 0.5
    store["x3"], store["c3"] = x3.numpy(), c3.numpy()
    store["channel_poly_d4v3"] = CPL(degree=4, num_variables=3)(x3, c3).numpy()
    # the full per-pixel path of the live model, spatial (infer.py:22-23: Deg4MobilePolyLayer) and non-spatial
    img = torch.rand(B, 3, H, W, generator=g)
    img8 = torch.randint(0, 256, (B, 3, H, W), generator=g).float() / 255
    store["img"], store["img8"] = img.numpy(), img8.numpy()
    for scale_name, scale in (("s02", 0.2), ("s1", 1.0)):
        coeffs = torch.randn(B, 3, 3, 126, generator=g) * scale
        store[f"{scale_name}_coeffs"] = coeffs.numpy()
        R, L, Hh = coeffs[:, 0], coeffs[:, 1], coeffs[:, 2]   # model.py:526
        tri = make_tri(ns, True, D4())
        for nm, x in (("img", img), ("img8", img8)):
            res = tri.generate_residual(x, R, L, Hh)
            store[f"{scale_name}_{nm}_residual"] = res.numpy()
            store[f"{scale_name}_{nm}_image"] = tri.generate_image(x, res).numpy()
        tri_c = make_tri(ns, True, CPL(degree=4, num_variables=5, num_out=3))
        store[f"{scale_name}_img_residual_channelpoly"] = tri_c.generate_residual(img, R, L, Hh).numpy()
        c35 = torch.randn(B, 3, 3, 35, generator=g) * scale
        store[f"{scale_name}_coeffs35"] = c35.numpy()
        tri_n = make_tri(ns, False, CPL(degree=4, num_variables=3, num_out=3))
        store[f"{scale_name}_img_residual_nonspatial"] = tri_n.generate_residual(img, c35[:, 0], c35[:, 1], c35[:, 2]).numpy()
    np.savez_compressed(os.path.join(OUT, "poly.npz"), **store)
    print("poly.npz", os.path.getsize(os.path.join(OUT, "poly.npz")) // 1024, "KiB,", len(store), "arrays")


if __name__ == "__main__":
    main()
