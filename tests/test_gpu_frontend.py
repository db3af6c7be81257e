"""GPU tile front end vs numpy: ImageNorm statistics and tiles BIT-EXACT (the kernel
follows numpy's float32 reduction order), stitching, and the streamed whole-frame path."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend_ref, unet_oracle
from sequitr_amd.frontend import FrameTiler, segment_frames
from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from tests.util import assert_bit_exact

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("shape,dtype", [((2, 48, 40), np.float32), ((3, 100, 130), np.uint16), ((2, 96, 128), np.uint8),
                                         ((1, 300, 417), np.uint16), ((2, 128, 64), np.float32), ((1, 7, 5), np.uint8),
                                         ((1, 512, 512), np.uint16), ((1, 1200, 1600), np.uint16)])
def test_frame_stats_bit_exact_with_numpy(shape, dtype):
    rng = np.random.default_rng(shape[1])
    if dtype == np.float32:
        fr = (rng.standard_normal(shape) * 30 + 100).astype(np.float32)
    else:
        fr = rng.integers(0, np.iinfo(dtype).max // 3, shape).astype(dtype)
    T = min(shape[1], shape[2], 32)
    tl = FrameTiler(shape[1:], tile=T, margin=0, device="cuda:0")
    mean, std = tl.stats(dev(fr))
    for i in range(shape[0]):
        a = np.array(fr[i], dtype="float")[..., None].astype("float32")[..., 0]           # OctopusData + ImagePipe casts
        assert_bit_exact(mean[i].cpu().numpy(), np.mean(a), "mean %d" % i)
        assert_bit_exact(std[i].cpu().numpy(), np.std(a), "std %d" % i)


def test_reference_vector_norm_out():
    """img_in -> norm_out was produced by the reference's ImageNorm itself."""
    img = G["img_in"]
    tl = FrameTiler(img.shape, tile=32, margin=4, device="cuda:0")
    tiles = tl.tiles(dev(img[None])).cpu().numpy()
    ref = G["norm_out"][..., 0]
    k = 0
    for y in tl.oy:
        for x in tl.ox:
            assert_bit_exact(tiles[k, ..., 0], np.ascontiguousarray(ref[y:y + 32, x:x + 32]), "tile %d" % k)
            k += 1


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_tiles_and_stitch_vs_oracle(dtype):
    rng = np.random.default_rng(3)
    fr = rng.integers(0, 200, (2, 150, 210)).astype(dtype)
    tl = FrameTiler((150, 210), tile=64, margin=8, device="cuda:0")
    assert tl.TR == 3 and tl.TC == 5
    got = tl.tiles(dev(fr)).cpu().numpy()
    assert_bit_exact(got, frontend_ref.tiles(fr, tl.oy, tl.ox, 64), "normalised tiles")
    raw = tl.tiles(dev(fr), normalise=False).cpu().numpy()
    assert_bit_exact(raw, frontend_ref.tiles(fr, tl.oy, tl.ox, 64, normalise=False), "raw tiles")
    masks = rng.integers(0, 3, (2 * 15, 64, 64)).astype(np.uint8)
    st = tl.stitch(dev(masks)).cpu().numpy()
    assert np.array_equal(st, frontend_ref.stitch(masks, tl.oy, tl.ox, tl.ymap, tl.xmap, 150, 210))
    # a tiling of a frame stitched back is the frame itself
    ident = (np.arange(150)[:, None] * 7 + np.arange(210)[None]) % 251
    ident = np.broadcast_to(ident, (2, 150, 210)).astype(np.uint8)
    back = tl.stitch(tl.tiles(dev(ident), normalise=False).to(torch.uint8).reshape(-1, 64, 64).contiguous())
    assert np.array_equal(back.cpu().numpy(), ident)


def test_segment_frames_streams_and_matches_per_tile_oracle():
    """5 raw uint16 frames of 96x160 through a 64-pixel U-Net: the streamed double-buffered path equals
    normalise -> tile -> oracle forward -> stitch done step by step."""
    params = {"shape": (64, 64), "filters": (16, 32), "device": "cuda:0"}
    net = UNet2D(params, "infer")
    w = init_unet_weights(params, 4)
    net.load_state_dict(w)
    rng = np.random.default_rng(8)
    fr = rng.integers(100, 4000, (5, 96, 160)).astype(np.uint16)
    got = segment_frames(net, fr, tile=64, margin=8, frames_per_batch=2)
    tl = FrameTiler((96, 160), tile=64, margin=8, device="cuda:0")
    tiles = frontend_ref.tiles(fr, tl.oy, tl.ox, 64)
    ref_masks = unet_oracle.predict_mask(unet_oracle.unet_forward(tiles, w, params))
    ref = frontend_ref.stitch(ref_masks, tl.oy, tl.ox, tl.ymap, tl.xmap, 96, 160)
    assert got.shape == (5, 96, 160) and np.array_equal(got, ref)
    seen = []
    assert segment_frames(net, fr, tile=64, margin=8, frames_per_batch=4,
                          on_masks=lambda first, m: seen.append((first, m.cpu().numpy()))) is None
    assert [s[0] for s in seen] == [0, 4] and np.array_equal(np.concatenate([s[1] for s in seen]), ref)


def test_errors_are_loud():
    tl = FrameTiler((64, 64), tile=32, margin=0, device="cuda:0")
    with pytest.raises(Exception):
        tl.tiles(torch.zeros((1, 64, 64), dtype=torch.uint8))                 # CPU tensor
    with pytest.raises(ValueError):
        tl.tiles(torch.zeros((1, 64, 65), dtype=torch.uint8, device="cuda:0"))
    with pytest.raises(ValueError):
        FrameTiler((20, 64), tile=32)
