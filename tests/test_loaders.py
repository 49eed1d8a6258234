"""Host input pipeline: bs_yolo_amd.loaders against the letterbox oracle and the one-shot preprocess."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

DEV = "cuda:0"


def _write_images(tmp_path, shapes, exts, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    files, arrays = [], []
    for i, ((h, w), ext) in enumerate(zip(shapes, exts)):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        f = tmp_path / f"im{i:03d}.{ext}"
        Image.fromarray(rgb).save(f, quality=95) if ext == "jpg" else Image.fromarray(rgb).save(f)
        files.append(str(f))
        arrays.append(rgb[:, :, ::-1].copy())  # BGR, as cv2.imread would return it (lossless formats)
    return files, arrays


def test_loader_needs_a_gpu_and_rejects_videos(tmp_path):
    from bs_yolo_amd import loaders as HL
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            HL.LoadImagesPinned([np.zeros((8, 8, 3), np.uint8)], device="cuda:0")
    assert HL.imread_bgr(str(tmp_path / "missing.png")) is None
    files, arrays = _write_images(tmp_path, [(5, 7)], ["png"])
    assert np.array_equal(HL.imread_bgr(files[0]), arrays[0])


def test_decode_pool_fills_shared_arenas(tmp_path):
    """The forked decode workers write BGR pixels at the given offsets of the shared arenas (no GPU involved)."""
    from bs_yolo_amd import loaders as HL
    shapes = [(48, 64), (33, 17), (5, 7), (64, 64), (10, 300)]
    files, arrays = _write_images(tmp_path, shapes, ["png", "bmp", "png", "jpg", "png"])
    arrays[3] = HL.imread_bgr(files[3])  # lossy format: the pool must equal the in-process decode
    assert [HL.image_size(f) for f in files] == shapes and HL.image_size(str(tmp_path / "nope.png")) is None
    pool = HL._DecodePool(3, 2, 1 << 20)
    try:
        cell = 64 * 64 * 3
        for slot in (0, 1, 0):
            res = pool.decode([(f, slot, k * cell, cell) for k, f in enumerate(files)])
            assert res == shapes
            host = pool.tensors[slot].numpy()
            for k, a in enumerate(arrays):
                assert np.array_equal(host[k * cell:k * cell + a.size].reshape(a.shape), a)
            host[:5 * cell] = 0
        res = pool.decode([(files[0], 0, 0, 100), (str(tmp_path / "nope.png"), 0, 0, cell)])
        assert isinstance(res[0], str) and res[1] is None  # too large for the cell: reported; unreadable: None (skipped)
        assert not pool.tensors[0].numpy()[:cell].any()
    finally:
        pool.close()
    assert all(not p.is_alive() for p, _ in pool.procs) and pool.procs == []


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
def test_loader_batches_equal_oracle_letterbox(tmp_path, half):
    """Mixed shapes (square letterbox) and equal shapes (minimal rectangle, LetterBox auto) through decode -> pinned arena ->
    upload -> letterbox: bit-identical to the oracle's preprocess of the same BGR arrays; order, paths, info strings and the
    short last batch as LoadImagesAndVideos yields them."""
    from bs_yolo_amd import loaders as HL
    from oracle import letterbox_ref as LB
    shapes = [(480, 640), (1080, 810), (333, 500), (640, 640), (97, 1001), (720, 1280), (500, 333)]
    files, arrays = _write_images(tmp_path, shapes, ["png", "bmp", "png", "bmp", "png", "png", "bmp"])
    loader = HL.LoadImagesPinned(str(tmp_path), batch=3, imgsz=640, half=half, device=DEV, workers=4, depth=2,
                                 decode="process" if half else "thread")
    assert len(loader) == 3 and loader.nf == 7
    seen = 0
    for batch in loader:
        n = len(batch.paths)
        assert batch.paths == files[seen:seen + n] and batch.orig_shapes == shapes[seen:seen + n]
        assert batch.info[0] == f"image {seen + 1}/7 {files[seen]}: "
        ref = LB.preprocess(arrays[seen:seen + n], (640, 640), half=half, pt=True, stride=32)
        assert batch.im.dtype == ref.dtype and torch.equal(batch.im.cpu(), ref)
        seen += n
    assert seen == 7
    assert sum(len(b.paths) for b in loader) == 7  # a second pass reuses the decode processes
    loader.close()
    # equal shapes -> auto (rect) letterbox: 480 x 640 stays 480 x 640
    same = [np.ascontiguousarray(a) for a in np.random.default_rng(2).integers(0, 256, (4, 480, 600, 3), dtype=np.uint8)]
    loader = HL.LoadImagesPinned(same, batch=4, imgsz=640, half=half, device=DEV)
    (batch,) = list(loader)
    ref = LB.preprocess(same, (640, 640), half=half, pt=True, stride=32)
    assert tuple(batch.im.shape) == tuple(ref.shape) == (4, 3, 512, 640) and torch.equal(batch.im.cpu(), ref)


@pytest.mark.gpu
def test_loader_many_batches_pipeline_reuse(tmp_path):
    """More batches than slots (depth 2): every arena is reused several times while earlier outputs are still alive."""
    from bs_yolo_amd import loaders as HL
    from bs_yolo_amd import letterbox as HLB
    rng = np.random.default_rng(5)
    ims = [rng.integers(0, 256, (int(rng.integers(60, 300)), int(rng.integers(60, 300)), 3), dtype=np.uint8) for _ in range(41)]
    outs = [b.im for b in HL.LoadImagesPinned(ims, batch=4, imgsz=(256, 320), device=DEV, workers=3, depth=2)]
    torch.cuda.synchronize()
    assert len(outs) == 11 and outs[-1].shape[0] == 1
    for k, o in enumerate(outs):
        ref = HLB.preprocess(ims[4 * k:4 * k + 4], (256, 320), half=True, device=DEV)
        assert torch.equal(o, ref)


@pytest.mark.gpu
def test_predict_stream_matches_unpipelined_path(tmp_path):
    from bs_yolo_amd import letterbox as HLB, loaders as HL, nms as HN
    from bs_yolo_amd.engine import YoloEngine
    from bs_yolo_amd.graphs import stock_cfg
    from oracle import yolo_ref as R
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, 0)
    for k in P:
        if ".cv3." in k and k.endswith(".2.bias"):
            P[k] = P[k] + 2.0
    shapes = [(300, 400), (480, 640), (200, 200), (640, 480), (123, 456)]
    files, arrays = _write_images(tmp_path, shapes, ["png"] * 5, seed=9)
    eng = YoloEngine(stock_cfg("yolo11", "n"), P, autotune=False)
    got = list(HL.predict_stream(eng, HL.LoadImagesPinned(files, batch=2, imgsz=320, device=DEV), conf=0.25, iou=0.7))
    assert [len(b.paths) for b, _, _ in got] == [2, 2, 1]
    i = 0
    for batch, det, counts in got:
        n = len(batch.paths)
        x = HLB.preprocess(arrays[i:i + n], (320, 320), half=True, device=DEV)
        y, _ = eng(x, want_raw=False)
        d, c = HN.nms_batched(y, 0.25, 0.7, max_det=300)
        HN.scale_boxes_batched(d, c, x.shape[2:], shapes[i:i + n])
        assert torch.equal(c, counts) and torch.equal(d, det) and int(c.sum()) > 0
        i += n
    eng.close()
