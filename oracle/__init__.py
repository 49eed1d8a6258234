"""CPU oracle for the YOLO detection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bs_yolo_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker (never as the thing measured or shipped).

Pinning status:
  * graph / Conv / C3k2 / C2f / SPPF / C2PSA / Detect / Segment / fuse / decode: PINNED against
    outputs of the reference itself (``tests/golden/*.npz`` produced by
    ``tests/golden/make_fixtures.py`` importing ``/root/reference`` in the build container).
  * non_max_suppression wrapper logic (ops.py:167-316): pinned against the reference function run
    with the oracle's greedy ``nms`` injected for the absent ``torchvision.ops.nms``;
    the greedy IoU kernel itself (torchvision, un-vendored, unpinned version) is a restatement of
    the published algorithm -> "parity unpinned" for tie-breaking.
  * LetterBox geometry (ratio / pad / output shape) pinned against the reference class;
    the pixel arithmetic of ``cv2.resize(INTER_LINEAR)`` (opencv-python, un-vendored) is a
    restatement of OpenCV's published fixed-point algorithm -> "parity unpinned" for pixels.
"""
