"""MC-sample inference on the HIP hot path (prob_utils/my_predictions/punet_predictions.py:15-136).

Compute lives in `mc_prediction` / `predict_image` / `pseudo_predict_image` (numpy image in, numpy out); the path-based
wrappers `punet_prediction` / `punet_pseudo_prediction` keep the reference signatures and need `imageio` for TIFF I/O
(file I/O is out of the hot-path scope).

Tiling follows what the reference delegates to the un-vendored `torch_em.util.prediction.predict_with_halo`
(punet_predictions.py:41-49: block 384x384, halo 64x64 -> 512x512 outer tile) as RECALLED in SURVEY.md §8c: blocks
read as full (block + 2 * halo) windows, reflect padding where a window leaves the image, per-tile standardisation,
halo cropped from the output.  That third-party routine is not in /root/reference -> tiling parity is unpinned; per-tile arithmetic is the
pinned `_custom_punet_prediction` body (punet_predictions.py:29-33).
"""
from __future__ import annotations

import os
from glob import glob

import numpy as np
import torch


def standardize(raw: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """[recalled] torch_em.transform.raw.standardize: (x - mean) / (std + eps) over the whole array."""
    raw = raw.astype("float32")
    return (raw - raw.mean()) / (raw.std() + eps)


@torch.no_grad()
def mc_prediction(model, raw: torch.Tensor, prior_samples: int = 8, eps=None) -> torch.Tensor:
    """_custom_punet_prediction (punet_predictions.py:29-33): forward(raw, None, False), then the mean of
    `prior_samples` sigmoid(sample(testing=True)) - one fused K7 launch instead of N Fcomb passes."""
    if hasattr(model, "pseudo_labels"):  # one call (the native runtime where it applies)
        y, _ = model.pseudo_labels(raw, prior_samples, eps=eps)
    else:  # any object with the reference's forward() / sample_consensus() surface
        model.forward(raw, None, training=False)
        y, _ = model.sample_consensus(prior_samples, eps=eps)
    return y


def _blocks(shape, block_shape):
    """Row-major blocks (y0, y1, x0, x1) of the image, the last ones clipped."""
    for y0 in range(0, shape[0], block_shape[0]):
        for x0 in range(0, shape[1], block_shape[1]):
            yield y0, min(y0 + block_shape[0], shape[0]), x0, min(x0 + block_shape[1], shape[1])


@torch.no_grad()
def predict_image(model, image: np.ndarray, prior_samples: int = 8, block_shape=(384, 384), halo=(64, 64),
                  tile_batch: int = 4, multiple_of: int = 8, shard_over_ranks: bool = True, eps=None) -> np.ndarray:
    """Tiled MC prediction of one 2-D image -> float32 mean-probability map (punet_predictions.py:35-50).
    Every block is read as the full (block + 2 * halo) window - what lies outside the image is reflect-padded, also for
    the clipped last blocks ([recalled] predict_with_halo) -, standardised per tile, predicted, and its halo cropped.
    All tiles have one shape and are batched `tile_batch` per launch; every tile gets its own latent draws
    (eps: optional [n_tiles, prior_samples, latent_dim] in row-major block order, for parity tests).
    Under an initialised `torch.distributed` group (one process per GPU) the tile batches are dealt round-robin to the
    ranks - tiles are independent, so this is replicas-only - and the disjoint partial maps are joined by
    ONE all_reduce(sum) of the output; every rank returns the full map."""
    assert image.ndim == 2, "2-D single-channel images (every reference script uses input_channels=1)"
    H, W = image.shape
    K = getattr(model, "num_classes", 1)
    out = np.zeros((H, W) if K == 1 else (K, H, W), dtype="float32")  # one map per class for num_classes > 1
    oh = -(-(block_shape[0] + 2 * halo[0]) // multiple_of) * multiple_of  # outer tile (512 x 512 for 384 / 64)
    ow = -(-(block_shape[1] + 2 * halo[1]) // multiple_of) * multiple_of
    img = image.astype("float32")

    def window(y0, x0):
        """The (oh, ow) window with origin (y0 - halo, x0 - halo): clipped to the image, the rest reflect-padded from the
        clipped part (per block, as predict_with_halo does - not from the whole image)."""
        ys, ye, xs, xe = y0 - halo[0], y0 - halo[0] + oh, x0 - halo[1], x0 - halo[1] + ow
        data = img[max(ys, 0):min(ye, H), max(xs, 0):min(xe, W)]
        pw = ((max(-ys, 0), max(ye - H, 0)), (max(-xs, 0), max(xe - W, 0)))
        return np.pad(data, pw, mode="reflect") if (pw[0][0] or pw[0][1] or pw[1][0] or pw[1][1]) else data

    items = [((y0, y1, x0, x1), standardize(window(y0, x0))) for (y0, y1, x0, x1) in _blocks((H, W), block_shape)]
    dev = model.device
    import torch.distributed as dist
    world = dist.get_world_size() if shard_over_ranks and dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    for n_chunk, i in enumerate(range(0, len(items), tile_batch)):
        if n_chunk % world != rank:
            continue
        chunk = items[i:i + tile_batch]
        raw = torch.from_numpy(np.stack([t for _, t in chunk])[:, None]).to(dev)
        e = None if eps is None else torch.as_tensor(eps)[i:i + len(chunk)].transpose(0, 1).contiguous()  # [N, tiles, L]
        pred = mc_prediction(model, raw, prior_samples, eps=e).cpu().numpy()
        pred = pred[:, 0] if K == 1 else pred
        for k, ((y0, y1, x0, x1), _) in enumerate(chunk):
            out[..., y0:y1, x0:x1] = pred[k][..., halo[0]:halo[0] + (y1 - y0), halo[1]:halo[1] + (x1 - x0)]
    if world > 1:  # every pixel was written by exactly one rank, the others hold 0 there
        joined = torch.from_numpy(out).to(dev)
        dist.all_reduce(joined, op=dist.ReduceOp.SUM)
        out = joined.cpu().numpy()
    return out


@torch.no_grad()
def pseudo_predict_image(model, image: np.ndarray, prior_samples: int = 8, upper_threshold=0.9, lower_threshold=0.1,
                         eps=None):
    """Whole-image forward, mean of N sigmoids and the binary consensus mask (punet_predictions.py:96-124):
    returns (mean float32 [H, W], consensus uint8 [H, W])."""
    patch = torch.from_numpy(standardize(image))[None, None].to(model.device)
    if hasattr(model, "pseudo_labels"):
        y, z = model.pseudo_labels(patch, prior_samples, upper_threshold, lower_threshold, masking=True, eps=eps)
    else:
        model.forward(patch, None, training=False)
        y, z = model.sample_consensus(prior_samples, upper_threshold, lower_threshold, masking=True, eps=eps)
    if getattr(model, "num_classes", 1) > 1:
        return y[0].cpu().numpy(), z[0].cpu().numpy().astype("uint8")
    return y[0, 0].cpu().numpy(), z[0, 0].cpu().numpy().astype("uint8")


def _imageio():
    try:
        import imageio.v3 as imageio
        return imageio
    except ImportError as e:  # pragma: no cover
        raise ImportError("punet_prediction's TIFF I/O needs `imageio` (not installed here); use predict_image() / "
                          "pseudo_predict_image() on arrays instead") from e


def punet_prediction(input_image_path: str, output_pred_path: str, model, prior_samples: int = 8, device: str = "cuda",
                     mysig=None):
    """Reference signature (punet_predictions.py:15-63)."""
    imageio = _imageio()
    model.eval()
    os.makedirs(output_pred_path, exist_ok=True)
    for img_path in glob(input_image_path):
        pred = predict_image(model, imageio.imread(img_path), prior_samples)
        out = os.path.join(output_pred_path, f"{os.path.basename(img_path)[:-4]}.tif")
        imageio.imwrite(out, pred, compression="zlib")
        print(f"Saved image at '{out}")


def punet_pseudo_prediction(input_image_path, output_pred_path, model, prior_samples=8, device="cuda", cellname_=None,
                            split_name: str = None):
    """Reference signature (punet_predictions.py:66-136)."""
    imageio = _imageio()
    os.makedirs(output_pred_path, exist_ok=True)
    model.eval()
    for i in glob(input_image_path + f"{cellname_}*.tif"):
        name = i.split("/")[-1]
        mypred, consensus = pseudo_predict_image(model, imageio.imread(i), prior_samples)
        dir1 = output_pred_path + f"annotations/{split_name}/{cellname_}/"
        dir2 = output_pred_path + f"consensus/{split_name}/{cellname_}/"
        os.makedirs(dir1, exist_ok=True)
        os.makedirs(dir2, exist_ok=True)
        imageio.imwrite(dir1 + name, mypred)
        imageio.imwrite(dir2 + name, consensus)
        print(f"{name}'s predictions saved")


# ---- tiled prediction of a plain model (prob_utils/my_predictions/unet_predictions.py:12-43) ---------------------------------------
class _PlainModelAsMC:
    """Adapter that lets predict_image() drive any callable model: one forward per tile batch, no latent draws."""

    def __init__(self, model):
        self.model = model
        self.device = getattr(model, "device", None) or next(model.parameters()).device
        self.num_classes = getattr(model, "num_classes", 1)

    def pseudo_labels(self, raw, n_samples, *a, eps=None, **k):
        return self.model(raw), None


def unet_predict_image(model, image: np.ndarray, block_shape=(384, 384), halo=(64, 64), tile_batch: int = 4) -> np.ndarray:
    """predict_with_halo(input_, model, block_shape=(384, 384), halo=(64, 64)) of unet_predictions.py:24-31 for a model that
    maps a standardised [B, 1, h, w] tile to [B, K, h, w] (this package's `Unet`, my_models/unet.py): same blocking, reflect
    padding, per-tile standardisation and halo crop as predict_image; returns the model's raw output map(s), squeezed."""
    return predict_image(_PlainModelAsMC(model), image, prior_samples=1, block_shape=block_shape, halo=halo, tile_batch=tile_batch)


@torch.no_grad()
def unet_prediction(input_path: str, output_path: str, model, device: str = "cuda"):
    """Reference signature (unet_predictions.py:12-43); the reference drives torch_em's UNet2d through it (third-party, out of
    scope) - here any model with the `Unet` forward surface."""
    imageio = _imageio()
    model.eval()
    os.makedirs(output_path, exist_ok=True)
    for img_path in glob(input_path):
        pred = unet_predict_image(model, imageio.imread(img_path)).squeeze()
        out = os.path.join(output_path, f"{os.path.basename(img_path)[:-4]}.tif")
        imageio.imwrite(out, pred, compression="zlib")
        print(f"Saved image at '{out}'")
