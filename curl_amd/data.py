"""Drop-in for the reference's data.py (the Adobe-5k-DPE style folder dataset of main.py:196-218) without
torchvision: same helper functions, same `Dataset(data_dict, normaliser, is_train, crop_h, crop_w)` and the same items
`{'input_img', 'output_img', 'mask', 'name'}` (float32 CHW in [0,1], mask [1,H,W] bool).

The transforms the reference takes from torchvision (data.py:103-117,155-169) are written with plain torch ops:
RandomCrop(pad_if_needed, fill 0) / CenterCrop, RandomHorizontalFlip, RandomVerticalFlip, RandomRotation(180,
nearest, fill 0) -- applied to the channel-stacked [input, output, mask] tensor so that all three see the same
transform.  This is host-side data loading (DataLoader workers); nothing here touches the GPU.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F


def get_data_ids(img_ids_filepath):
    """data.py:31-40: one integer id per line."""
    with open(img_ids_filepath) as f:
        return [int(x.rstrip()) for x in f.readlines() if x.strip() and not x.startswith('.')]


def get_data_dict(data_dirpath):
    """data.py:43-72: <dir>/*input*/, <dir>/*output*/, <dir>/*mask*/ with identical file names `<id>.<ext>`."""
    data_dirs = sorted(os.listdir(data_dirpath))
    try:
        input_dir = [d for d in data_dirs if 'input' in d][0]
        output_dir = [d for d in data_dirs if 'output' in d][0]
        mask_dir = [d for d in data_dirs if 'mask' in d][0]
    except IndexError:
        raise OSError("{} must contain a directories containing the words 'input', 'output' respectively".format(data_dirpath))
    full = {k: os.path.join(data_dirpath, d) for k, d in (("input_img", input_dir), ("output_img", output_dir), ("mask", mask_dir))}
    names = {k: [f for f in sorted(os.listdir(p)) if not f.startswith('.')] for k, p in full.items()}
    assert names["input_img"] == names["output_img"], "Input and output image directories should have the same file names."
    assert names["input_img"] == names["mask"], "Input image and mask directories should have the same file names."
    return {int(os.path.splitext(fn)[0]): {k: os.path.join(full[k], fn) for k in full} for fn in names["input_img"]}


def filter_data_dict(data_dict, image_id_list):
    """data.py:75-80."""
    return {new_idx: data_dict[idx] for new_idx, idx in enumerate(image_id_list)}


def _rotate_nearest(x, degrees):
    """RandomRotation's kernel: rotate [C,H,W] about the centre, nearest neighbour, zeros outside, same size."""
    a = math.radians(degrees)
    _, H, W = x.shape
    # output pixel -> input pixel (inverse rotation) in normalised coordinates, aspect-corrected
    cos, sin = math.cos(a), math.sin(a)
    theta = torch.tensor([[cos, sin * H / W, 0.0], [-sin * W / H, cos, 0.0]], dtype=torch.float32)
    grid = F.affine_grid(theta[None], (1, x.shape[0], H, W), align_corners=False)
    return F.grid_sample(x[None], grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]


class Dataset(torch.utils.data.Dataset):
    def __init__(self, data_dict, normaliser=2 ** 8 - 1, is_train=False, crop_h=256, crop_w=256, seed=None):
        self.data_dict = data_dict
        self.normaliser = normaliser
        self.is_train = is_train
        self.crop_h, self.crop_w = crop_h, crop_w
        self.rng = torch.Generator()
        self.seed = seed
        if seed is not None:
            self.rng.manual_seed(seed)
        self._worker_seed = None

    def __len__(self):
        return len(self.data_dict.keys())

    @staticmethod
    def load_image(img_filepath, normaliser, mono=False):
        """data.py:127-139."""
        from PIL import Image
        img = Image.open(img_filepath)
        img = img.convert('1') if mono else img
        return Dataset.normalise_image(np.array(img), normaliser)

    @staticmethod
    def normalise_image(img, normaliser):
        return img.astype('float32') / normaliser

    def _rand(self):
        # Every DataLoader worker starts from a COPY of this object: with a private generator alone all workers would
        # draw the same crop / flip / rotation stream, and every epoch (workers are recreated from the parent's
        # never-advanced copy) would replay it.  The reference draws from the global RNG, which DataLoader reseeds per
        # worker and per epoch (base_seed + worker id); the private generator is re-seeded from that same seed.
        info = torch.utils.data.get_worker_info()
        if info is not None and self._worker_seed != info.seed:
            self.rng.manual_seed((info.seed + 0x9E3779B97F4A7C15 * ((self.seed or 0) + 1)) % (1 << 63))
            self._worker_seed = info.seed
        return float(torch.rand((), generator=self.rng))

    def _crop(self, x):
        _, H, W = x.shape
        ch, cw = self.crop_h, self.crop_w
        if self.is_train:  # RandomCrop(pad_if_needed=True, fill=0): pad both sides of a too-small axis, then a random window
            if W < cw:
                x = F.pad(x, (cw - W, cw - W, 0, 0))
            if H < ch:
                x = F.pad(x, (0, 0, ch - H, ch - H))
            _, H, W = x.shape
            top = int(self._rand() * (H - ch + 1)) if H > ch else 0
            left = int(self._rand() * (W - cw + 1)) if W > cw else 0
        else:              # CenterCrop: pad symmetrically if smaller, then the centred window
            if H < ch or W < cw:
                ph, pw = max(ch - H, 0), max(cw - W, 0)
                x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
                _, H, W = x.shape
            top, left = int(round((H - ch) / 2.0)), int(round((W - cw) / 2.0))
        return x[:, top:top + ch, left:left + cw]

    def transform(self, input_img, output_img, mask):
        """data.py:152-174: one transform for the stacked [input, output, mask]."""
        if input_img.ndim == 2:
            input_img, output_img = input_img[..., None].repeat(3, 2), output_img[..., None].repeat(3, 2)
        stack = np.concatenate([input_img[..., :3], output_img[..., :3], mask.reshape(mask.shape[0], mask.shape[1], 1)], axis=2)
        if self.normaliser == 1:   # data.py:158-159: raw 8-bit values -> to_tensor divides by 255
            x = torch.from_numpy(np.ascontiguousarray(stack.astype(np.uint8))).permute(2, 0, 1).float().div(255)
        else:
            x = torch.from_numpy(np.ascontiguousarray(stack)).permute(2, 0, 1).float()
        x = self._crop(x)
        if self.is_train:
            if self._rand() < 0.5:
                x = x.flip(2)
            if self._rand() < 0.5:
                x = x.flip(1)
            x = _rotate_nearest(x, (self._rand() * 2.0 - 1.0) * 180.0)
        return x[:3].contiguous(), x[3:6].contiguous(), x[6:7].contiguous()

    def __getitem__(self, idx):
        e = self.data_dict[idx]
        input_img = Dataset.load_image(e['input_img'], normaliser=self.normaliser)
        output_img = Dataset.load_image(e['output_img'], normaliser=self.normaliser)
        mask = Dataset.load_image(e['mask'], normaliser=self.normaliser, mono=True)
        input_img, output_img, mask = self.transform(input_img, output_img, mask)
        return {'input_img': input_img, 'output_img': output_img, 'mask': mask > 0,
                'name': e['input_img'].split("/")[-1]}
