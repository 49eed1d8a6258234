"""Host input pipeline for the predict path: image files -> pinned staging -> HBM -> letterboxed model input.

Mirror of the image branch of ``LoadImagesAndVideos`` (data/loaders.py:284-449: sorted file list, batches of `batch`
images, ``paths, imgs, info`` per batch) fused with ``BasePredictor.pre_transform`` / ``preprocess``
(engine/predictor.py:116-161: LetterBox with ``auto`` = all images of the batch share one shape, stack, BGR->RGB,
HWC->CHW, to device, half, /255).  The reference does those steps one after another on the host, per batch, between two
forwards; here they overlap with the GPU:

  * decode      -- worker PROCESSES (forked once per loader; PIL's JPEG path holds the GIL for most of a decode, threads
                   scale to ~1.2x) write straight into ONE pinned, shared uint8 arena per in-flight batch, images packed back
                   to back in BGR order (what ``cv2.imread`` hands the reference); `decode="thread"` keeps everything in
                   one process;
  * upload      -- one asynchronous copy of the arena per batch on a dedicated HIP stream (no per-image copies);
  * letterbox   -- ``bsy_letterbox`` on the consumer's stream, ordered after the upload by an event: resize (OpenCV's
                   8-bit bilinear, restated), pad 114, channel swap, /255 -> (B, 3, H, W) fp16/fp32;
  * `depth` batches are in flight, so batch k+1 is decoded and uploaded while batch k runs through the engine.

Videos / streams / screenshots stay on the reference loaders (they need OpenCV's capture classes).  JPEG decoding is
PIL's libjpeg here and OpenCV's in the reference: the two may differ by +-1 on some pixels (unpinned third-party
arithmetic, cv2 is not in this image); lossless formats (PNG, BMP, TIFF) decode identically.  No CPU fallback: the
loader needs a GPU.
"""
from __future__ import annotations

import ctypes as C
import gc
import math
import mmap
import multiprocessing as mp
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import lib as L
from .letterbox import LetterBox

IMG_FORMATS = {"bmp", "dng", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp", "pfm"}  # data/utils.py:39 (HEIC: needs pillow-heif)


def imread_bgr(path: str) -> Optional[np.ndarray]:
    """``cv2.imread(path)`` semantics through PIL: (h, w, 3) uint8 BGR, None when the file cannot be read."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")
            w, h = im.size
            # the raw encoder writes B, G, R in C (1.9 ms for a 640 x 480 JPEG incl. the decode; a numpy channel flip of the
            # decoded array costs another 2.5 ms)
            return np.frombuffer(im.tobytes("raw", "BGR"), dtype=np.uint8).reshape(h, w, 3)
    except Exception:
        return None


def image_size(path: str):
    """(h, w) from the file header only (no decode), None when unreadable."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            w, h = im.size
        return h, w
    except Exception:
        return None


def _worker_main(conn, arenas):
    """Decode worker (forked child): never touches the GPU.  Jobs: (path, slot, offset, capacity) -> BGR pixels written at
    arenas[slot][offset : offset + h*w*3]; reply (h, w), None (unreadable file) or an error text (does not fit)."""
    gc.disable()  # no finalizers of inherited GPU objects in this process
    for fd in os.listdir("/proc/self/fd"):  # drop the inherited GPU device handles: this process is not a GPU client
        try:
            if os.readlink(f"/proc/self/fd/{fd}").startswith(("/dev/kfd", "/dev/dri")):
                os.close(int(fd))
        except OSError:
            pass
    views = [np.frombuffer(a, dtype=np.uint8) for a in arenas]
    try:
        while True:
            job = conn.recv()
            if job is None:
                break
            path, slot, off, cap = job
            im = imread_bgr(path)
            if im is None:
                conn.send(None)  # unreadable: the loader skips it (data/loaders.py:427)
            elif im.size > cap:
                conn.send(f"{path}: {im.shape[0]} x {im.shape[1]} image needs {im.size} bytes, the arena cell holds {cap}")
            else:
                views[slot][off:off + im.size] = im.reshape(-1)
                conn.send((int(im.shape[0]), int(im.shape[1])))
    except (EOFError, KeyboardInterrupt):
        pass
    finally:
        os._exit(0)  # skip interpreter teardown (inherited HIP state must not run its exit handlers here)


class _DecodePool:
    """`n` forked decode processes + `depth` shared anonymous arenas (MAP_SHARED mmaps made BEFORE the fork, pinned by
    the parent AFTER it so the registration is not inherited)."""

    def __init__(self, n: int, depth: int, arena_bytes: int):
        self.arenas = [mmap.mmap(-1, arena_bytes) for _ in range(depth)]
        ctx = mp.get_context("fork")
        self.procs, self.idle = [], queue.Queue()
        for _ in range(n):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(b, self.arenas), daemon=True)
            p.start()
            b.close()
            self.procs.append((p, a))
            self.idle.put(a)
        self.tensors = [torch.frombuffer(a, dtype=torch.uint8) for a in self.arenas]
        self.registered = []
        if torch.cuda.is_available():  # page-lock the arenas: asynchronous uploads straight from where the workers write
            for t in self.tensors:
                if int(torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)) == 0:
                    self.registered.append(t.data_ptr())
        self.threads = ThreadPoolExecutor(n)

    def decode(self, jobs):
        """jobs: [(path, slot, offset, capacity)] -> [(h, w) | None | error text], in order (blocks; the pipes release the GIL)."""
        def one(job):
            c = self.idle.get()
            try:
                c.send(job)
                return c.recv()
            finally:
                self.idle.put(c)
        return list(self.threads.map(one, jobs))

    def close(self):
        for p, c in self.procs:
            try:
                c.send(None)
                c.close()
            except Exception:
                pass
        for p, _ in self.procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        self.procs = []
        self.threads.shutdown(wait=False)
        for ptr in self.registered:
            torch.cuda.cudart().cudaHostUnregister(ptr)
        self.registered, self.tensors = [], []


class Batch:
    """One loader batch: `im` (B, 3, H, W) device tensor ready for the engine + the reference's (paths, info) and what
    scale_boxes needs (orig_shapes)."""

    def __init__(self, im, paths, info, orig_shapes, im0s):
        self.im, self.paths, self.info, self.orig_shapes, self.im0s = im, paths, info, orig_shapes, im0s

    def __iter__(self):  # `for paths, im0s, info in loader` as with the reference loader
        return iter((self.paths, self.im0s, self.info))


class LoadImagesPinned:
    def __init__(self, path: Union[str, Sequence], batch: int = 1, imgsz=640, stride: int = 32, half: bool = True,
                 device="cuda:0", workers: int = 8, depth: int = 2, pt: bool = True, keep_im0: bool = False,
                 decode: str = "process", max_image_bytes: int = 3 << 20):
        """decode: "process" (forked workers + shared pinned arenas of `batch` cells of max_image_bytes each -- 3 MiB holds
        1280 x 720; a larger image raises) or "thread" (one process, arenas grow on demand).  Array sources are copied by
        the parent in either mode.  One iteration at a time per loader."""
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("bs_yolo_amd.loaders needs a ROCm GPU (no CPU fallback)")
        self.bs, self.stride, self.half, self.pt, self.keep_im0 = int(batch), int(stride), bool(half), bool(pt), bool(keep_im0)
        self.imgsz = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)
        self.workers, self.depth = max(1, int(workers)), max(1, int(depth))
        if decode not in ("process", "thread"):
            raise ValueError("decode must be 'process' or 'thread'")
        self.decode_mode, self.max_image_bytes, self._pool = decode, int(max_image_bytes), None
        # file list exactly as data/loaders.py:324-349 builds it (txt list, directories, globs are the caller's business
        # here: pass the expanded list; arrays are taken as already decoded BGR images)
        if isinstance(path, (str, Path)) and Path(path).suffix == ".txt":
            parent = Path(path).parent
            items = [str((parent / p).resolve()) if not Path(p).is_absolute() else p for p in Path(path).read_text().splitlines() if p.strip()]
        elif isinstance(path, (str, Path)):
            p = Path(path)
            items = sorted(str(f) for f in p.iterdir()) if p.is_dir() else [str(p)]
        else:
            items = list(path)
            if all(isinstance(p, (str, Path)) for p in items):
                items = sorted(str(p) for p in items)
        self.files = []
        for it in items:
            if isinstance(it, np.ndarray):
                self.files.append(it)
            elif str(it).rpartition(".")[-1].lower() in IMG_FORMATS:
                self.files.append(str(it))
            elif str(it).rpartition(".")[-1].lower() == "heic":
                raise NotImplementedError("HEIC needs pillow-heif; use the reference loader")
            else:
                raise NotImplementedError(f"{it}: videos / streams stay on the reference loaders (data/loaders.py)")
        self.ni = self.nf = len(self.files)
        if self.nf == 0:
            raise FileNotFoundError(f"No images found in {path}")
        self.mode = "image"

    def __len__(self):
        return math.ceil(self.nf / self.bs)

    # ---- producer ------------------------------------------------------------------------------------------------------
    def _decode(self, item):
        if isinstance(item, np.ndarray):
            if item.dtype != np.uint8 or item.ndim != 3 or item.shape[2] != 3:
                raise TypeError("array sources must be (h, w, 3) uint8 BGR")
            return item
        return imread_bgr(item)

    def _produce(self, q: "queue.Queue", free: "queue.Queue", stop: threading.Event):
        torch.cuda.set_device(self.device)
        copy_stream = torch.cuda.Stream(self.device)
        procs, cell = self._pool, self.max_image_bytes
        try:
            with ThreadPoolExecutor(self.workers) as pool:
                for b0 in range(0, self.nf, self.bs):
                    items = self.files[b0:b0 + self.bs]
                    slot = None
                    while slot is None and not stop.is_set():
                        try:
                            slot = free.get(timeout=0.1)
                        except queue.Empty:
                            pass
                    if stop.is_set():
                        return
                    if slot["consumed"] is not None:
                        slot["consumed"].synchronize()  # the letterbox that read this slot's device arena is done
                    if procs is not None:
                        # fixed cells of max_image_bytes: the workers decode straight into the pinned arena, no header pass
                        host = slot["pinned"].numpy()
                        jobs = [(it, slot["index"], k * cell, cell) for k, it in enumerate(items) if isinstance(it, str)]
                        res = iter(procs.decode(jobs))
                        metas, offs_all = [], [k * cell for k in range(len(items))]
                        for k, it in enumerate(items):
                            if isinstance(it, str):
                                r = next(res)
                                if isinstance(r, str):
                                    raise ValueError(r + ": raise max_image_bytes or use decode='thread'")
                                metas.append(r)
                            else:
                                im = self._decode(it)
                                if im.size > cell:
                                    raise ValueError(f"array source {k} needs {im.size} bytes, the arena cell holds {cell}")
                                np.copyto(host[k * cell:k * cell + im.size].reshape(im.shape), im)
                                metas.append(im.shape[:2])
                    else:
                        ims = list(pool.map(self._decode, items))
                        metas = [None if im is None else im.shape[:2] for im in ims]
                        offs_all, total = [], 0
                        for m in metas:
                            offs_all.append(total)
                            total += 0 if m is None else (m[0] * m[1] * 3 + 255) // 256 * 256
                        if slot["pinned"] is None or slot["pinned"].numel() < total:
                            slot["pinned"] = torch.empty(int(total * 1.25), dtype=torch.uint8).pin_memory()
                        host = slot["pinned"].numpy()
                        list(pool.map(lambda a: np.copyto(host[a[0]:a[0] + a[1].size].reshape(a[1].shape), a[1]),
                                      [(o, im) for o, im in zip(offs_all, ims) if im is not None]))
                    paths, info, offs, shapes = [], [], [], []
                    for k, (it, hw) in enumerate(zip(items, metas)):
                        if hw is None:  # data/loaders.py:427 warns and skips
                            continue
                        name = it if isinstance(it, str) else f"image{b0 + k}.jpg"
                        paths.append(name)
                        info.append(f"image {b0 + k + 1}/{self.nf} {name}: ")
                        offs.append(offs_all[k])
                        shapes.append((int(hw[0]), int(hw[1])))
                    if not shapes:
                        free.put(slot)
                        continue
                    need = offs[-1] + shapes[-1][0] * shapes[-1][1] * 3
                    if slot["dev"] is None or slot["dev"].numel() < need:
                        slot["dev"] = torch.empty(len(items) * cell if procs is not None else int(need * 1.25), dtype=torch.uint8,
                                                  device=self.device)
                    with torch.cuda.stream(copy_stream):
                        if procs is not None:  # one copy per image: the cells are mostly empty
                            for o, (h, w) in zip(offs, shapes):
                                slot["dev"][o:o + h * w * 3].copy_(slot["pinned"][o:o + h * w * 3], non_blocking=True)
                        else:
                            slot["dev"][:need].copy_(slot["pinned"][:need], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    im0s = [host[o:o + h * w * 3].reshape(h, w, 3).copy() for o, (h, w) in zip(offs, shapes)] if self.keep_im0 else None
                    slot.update(uploaded=ev, offs=offs, shapes=shapes, paths=paths, info=info, im0s=im0s)
                    q.put(slot)
            q.put(None)
        except BaseException as e:  # surface decoder / allocation errors in the consumer
            q.put(e)

    # ---- consumer ------------------------------------------------------------------------------------------------------
    def __iter__(self):
        q: "queue.Queue" = queue.Queue()
        free: "queue.Queue" = queue.Queue()
        if self.decode_mode == "process" and self._pool is None and any(isinstance(f, str) for f in self.files):
            self._pool = _DecodePool(self.workers, self.depth, self.bs * self.max_image_bytes)
        for i in range(self.depth):
            free.put(dict(pinned=self._pool.tensors[i] if self._pool else None, dev=None, consumed=None, index=i))
        stop = threading.Event()
        th = threading.Thread(target=self._produce, args=(q, free, stop), daemon=True)
        th.start()
        try:
            while True:
                slot = q.get()
                if slot is None:
                    return
                if isinstance(slot, BaseException):
                    raise slot
                yield self._finish(slot)
                free.put(slot)
        finally:
            stop.set()

    def close(self):
        """Stop the decode processes and release the shared arenas (also runs when the loader is collected)."""
        if self._pool is not None:
            torch.cuda.synchronize(self.device)  # no upload still reads an arena
            self._pool.close()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _finish(self, slot) -> Batch:
        dev = self.device
        shapes = slot["shapes"]
        same = len(set(shapes)) == 1
        lb = LetterBox(self.imgsz, auto=same and self.pt, stride=self.stride)  # predictor.py:152-157
        geoms = [lb.geometry(s) for s in shapes]
        out_hw = {(g[0], g[1]) for g in geoms}
        if len(out_hw) != 1:
            raise ValueError(f"letterboxed images differ in shape: {out_hw}")
        H2, W2 = out_hw.pop()
        B = len(shapes)
        base = slot["dev"].data_ptr()
        ptrs = torch.tensor([base + o for o in slot["offs"]], dtype=torch.int64).to(dev)
        hw = torch.tensor([[s[0], s[1]] for s in shapes], dtype=torch.int32).to(dev)
        geom = torch.tensor([[g[2], g[3], g[4], g[5]] for g in geoms], dtype=torch.int32).to(dev)
        out = torch.empty((B, 3, H2, W2), dtype=torch.float16 if self.half else torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(slot["uploaded"])
        L.check(L.lib.bsy_letterbox(C.c_void_p(ptrs.data_ptr()), C.c_void_p(hw.data_ptr()), C.c_void_p(geom.data_ptr()), B,
                                    H2, W2, C.c_void_p(out.data_ptr()), L.dtype_code(out.dtype), C.c_void_p(cur.cuda_stream)))
        done = torch.cuda.Event()
        done.record(cur)
        slot["consumed"] = done
        out._bsy_keepalive = (ptrs, hw, geom)
        return Batch(out, slot["paths"], slot["info"], [tuple(s) for s in shapes], slot["im0s"])


def predict_stream(engine, loader: LoadImagesPinned, conf: float = 0.25, iou: float = 0.7, max_det: int = 300,
                   classes=None, agnostic: bool = False):
    """The predict loop of the hot path (engine/predictor.py:240-262 stream_inference; the reference builds its own Results from these tensors,
    models/yolo/detect/predict.py:34-45):
    for every loader batch -> (batch, det (B, max_det, 6) fp32 with boxes scaled back to each original image
    (scale_boxes, ops.py:92-127), counts (B,) int32), all device-resident; the next batch decodes and uploads meanwhile."""
    from .nms import nms_batched, scale_boxes_batched
    for batch in loader:
        y = engine(batch.im, want_raw=False)[0]
        det, counts = nms_batched(y, conf, iou, classes=classes, agnostic=agnostic, max_det=max_det)
        scale_boxes_batched(det, counts, batch.im.shape[2:], batch.orig_shapes)
        yield batch, det, counts
