"""Multi-GPU layout of the curve path: one process per GPU.

Default: images shard across ranks; every pixel depends only on its own three values and its own
image's 160 knots, so there is NO data-path collective.  Optional shared-encoder / split-pixels
layout: the knots ([B,160] float32, 640 B per image) are the only thing that crosses xGMI -- one
broadcast or all-gather over RCCL -- and each rank applies them to its row slab of every image.
"""
import torch
import torch.distributed as dist


def image_shard(n_images, rank, world):
    """Contiguous, balanced [start, stop) of the batch for `rank` (first n%world ranks get one extra)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_images, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def row_slab(height, rank, world):
    """Rows [r0, r1) of every image owned by `rank` in the split-pixels layout."""
    return image_shard(height, rank, world)


def broadcast_knots(knots, src=0, group=None):
    """Shared-encoder layout: rank `src` ran the encoder; everyone receives the raw knots [B,160].
    `knots` must be allocated with the right shape on every rank (contents ignored off `src`)."""
    if dist.is_available() and dist.is_initialized():  # a single-rank group still goes through the backend (RCCL)
        dist.broadcast(knots, src=src, group=group)
    return knots


def allgather_knots(local_knots, group=None):
    """Each rank encoded its image shard ([b_local,160]); returns the knots of the whole batch in rank order.
    Shards may differ by one row (image_shard); they are padded to the longest for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_knots
    world = dist.get_world_size(group)
    n_local = torch.tensor([local_knots.shape[0]], device=local_knots.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    padded = local_knots.new_zeros((longest, local_knots.shape[1]))
    padded[: local_knots.shape[0]] = local_knots
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def apply_row_slab(layer, img, mask, L, R, H, rank, world, out=None):
    """Split-pixels layout: apply the curve layer to this rank's rows of every image.  Returns (out, reg, (r0, r1))
    with `out` a FULL-SIZE tensor (allocated here if not given) whose rows [r0, r1) hold the result; rows outside the
    slab are NOT written (uninitialised memory unless the caller passed its own `out`).  `reg` is the same on every
    rank, empty slabs included: it depends on the knots only.

    `layer` is a CURLLayer (its knot slicing is applied) or any callable (img, mask, L, R, H) -> (img, reg).
    * CURLLayer on a HIP device, nothing requiring a gradient: the rows are processed in place through the stride-aware
      entry point (ops.curl_layer_forward_rows -> curl_layer_fwd_slab_f32) -- no .contiguous() copy of the slab (a row
      slice of an NCHW tensor is three separate chunks per image), 24-28 B/px like the whole-image call.
    * anything else (another callable, CPU rehearsals, or an input that requires grad -- the slab entry point has no
      autograd node): the slice-and-copy route THROUGH `layer`, which stays differentiable."""
    from .model import CURLLayer
    r0, r1 = row_slab(img.shape[2], rank, world)
    if out is None:
        out = torch.empty_like(img)
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (img, L, R, H))
    if r1 == r0:  # more ranks than rows: no pixels here, but the regulariser is still owed (knots only): 1-row stand-in
        _, reg = layer(img[:, :, :1, :].contiguous(), None if mask is None else mask[:, :, :1, :].contiguous(), L, R, H)
        return out, reg, (r0, r1)
    if img.is_cuda and isinstance(layer, CURLLayer) and not needs_grad:
        from . import ops
        L, R, H = L[:, :layer.num_lab_points], R[:, :layer.num_rgb_points], H[:, :layer.num_hsv_points]  # model.py:153,159,165
        flags = (ops.F_PWL if layer.paper_pwl else 0) | layer.flags  # CURLLayer(paper_pwl=True): the non-parity option
        _, reg = ops.curl_layer_forward_rows(img, mask, L, R, H, (r0, r1), out, flags=flags)
        return out, reg, (r0, r1)
    sub = img[:, :, r0:r1, :].contiguous()
    sub_mask = None if mask is None else mask[:, :, r0:r1, :].contiguous()
    o, reg = layer(sub, sub_mask, L, R, H)
    out[:, :, r0:r1, :] = o  # (autograd records the slice copy: `out` joins the graph when `o` carries one)
    return out, reg, (r0, r1)


def apply_row_slab_trispace(img, coeffs, rank, world, out=None, residual_only=False):
    """The same layout for the polynomial model (model.py:499-520): rows [r0, r1) of every image through
    ops.trispace_forward_rows, whose pixel coordinates are the FULL image's (y = row / H with the image's height --
    slicing the tensor and calling the whole-image entry point would renumber the rows)."""
    from . import ops
    r0, r1 = row_slab(img.shape[2], rank, world)
    if out is None:
        out = torch.empty_like(img)
    if r1 > r0:
        ops.trispace_forward_rows(img, coeffs, (r0, r1), out, residual_only=residual_only)
    return out, (r0, r1)
