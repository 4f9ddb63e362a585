"""Single-image inference CLI for the curve model, with the flags of the reference's infer.py:14-17.

    python -m curl_amd.infer --img_path in.png --mask_path mask.png --model_file ckpt.pt --out_path out.png

Shape of the reference's infer.py:32-47, kept: the encoder sees a 320x320 resize+centre-crop of the image, the
curves are applied to the FULL-RESOLUTION image, the result is composited on white where the mask is 0 and saved.
Everything per-pixel runs on the GPU: the decoded uint8 HWC image goes up (3-4 B/px), is converted, enhanced,
composited and quantised on the device, and uint8 HWC comes back.
`--model_file random` builds a randomly initialised model (there is no curve-model checkpoint in the reference tree).
"""
import argparse

import numpy as np
import torch

from . import model as model_mod
from . import ops
from .convert_state import convert_state_dict


def build_net(model_file, device, arch="curl"):
    if arch == "trispace":  # infer.py:22-23
        net = model_mod.TriSpaceRegNet(polynomial_order=4, spatial=True, is_train=False,
                                       polylayer=model_mod.Deg4MobilePolyLayer())
    else:
        net = model_mod.GCURLNet(encoder_size=320)
    if model_file != "random":
        ckpt = torch.load(model_file, map_location="cpu")  # infer.py:25
        state = convert_state_dict(ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt)  # infer.py:28
        net.load_state_dict(state)
    return net.to(device).eval()


def encoder_view(img, mask, size=320):
    """Resize([320]) (short side) + CenterCrop(320) of infer.py:32-36, on the device."""
    _, _, H, W = img.shape
    scale = size / min(H, W)
    nh, nw = max(size, round(H * scale)), max(size, round(W * scale))
    small = torch.nn.functional.interpolate(img, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)
    msmall = torch.nn.functional.interpolate(mask, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)
    top, left = (nh - size) // 2, (nw - size) // 2
    return small[:, :, top:top + size, left:left + size], (msmall[:, :, top:top + size, left:left + size] > 0).float()


@torch.no_grad()
def enhance(net, img_u8, mask_u8, device):
    """img_u8: HxWx3|4 uint8, mask_u8: HxW uint8 ('L').  Returns HxWx3 uint8 (numpy).
    The full-resolution pass runs on the file's own bytes (ops.*_u8hwc: byte/255, the model's per-pixel part, the
    white background and the truncating *255 in one launch); only the 320x320 encoder view is made in float."""
    rgb = torch.from_numpy(np.ascontiguousarray(np.asarray(img_u8)[..., :3])).to(device)[None]  # alpha dropped
    white = torch.from_numpy(np.array(mask_u8, dtype=np.uint8)).to(device)[None]
    x = ops.u8hwc_to_f32chw(rgb)
    tmask = white[:, None].float() / 255.0  # to_tensor
    small, msmall = encoder_view(x, tmask)
    del x
    if isinstance(net, model_mod.TriSpaceRegNet):
        coeffs = torch.stack(net.generate_coefficients(small, msmall), 1)       # infer.py:44, model.py:522-527
        return ops.trispace_forward_u8hwc(rgb, coeffs, white)[0].cpu().numpy()  # infer.py:44-47
    knots = net.predict_knots(small * msmall)  # the encoder sees the masked 320x320 view
    L, R, H = knots[:, :net.curve_break_1], knots[:, net.curve_break_1:net.curve_break_2], knots[:, net.curve_break_2:]
    # the reference applies the mask only when compositing (infer.py:44-46): no layer mask
    cl = net.curllayer
    out, _ = ops.curl_layer_forward_u8hwc(rgb, None, L[:, :cl.num_lab_points].contiguous(),
                                          R[:, :cl.num_rgb_points].contiguous(),
                                          H[:, :cl.num_hsv_points].contiguous(), white)
    return out[0].cpu().numpy()


def infer(argv=None):
    parser = argparse.ArgumentParser(description="Run image enhancement model on a single image")
    parser.add_argument("--img_path", type=str, required=True, help="Path to image to enhancement")
    parser.add_argument("--mask_path", type=str, required=True, help="Path to image to enhancement")
    parser.add_argument("--model_file", type=str, required=True, help="Path to model checkpoint file ('random' = random init)")
    parser.add_argument("--out_path", type=str, required=True, help="Path to write output image to")
    parser.add_argument("--arch", choices=("trispace", "curl"), default="trispace",
                        help="trispace = the reference's infer.py model (TriSpaceRegNet + Deg4MobilePolyLayer); "
                             "curl = the curve model (GCURLNet)")
    args = parser.parse_args(argv)
    from PIL import Image
    device = torch.device("cuda:0")
    net = build_net(args.model_file, device, args.arch)
    img = np.asarray(Image.open(args.img_path))
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=2)
    mask = np.asarray(Image.open(args.mask_path).convert("L"))
    out = enhance(net, img, mask, device)
    Image.fromarray(out).save(args.out_path)


if __name__ == "__main__":
    infer()
