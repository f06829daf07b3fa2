"""A generated stereo mini-sequence in the on-disk layout the reference's stereo_kitti driver reads (SURVEY.md 8d,
config 1):  <seq>/image_02/%06d.png, <seq>/image_03/%06d.png (/root/reference/Examples/Stereo/stereo_kitti.cc:79-80,
213-225), timestamp.txt (:179-206), ObjectTracking.txt in KITTI-tracking label format (src/Tracking.cc:485-640),
poses.txt with 12 floats per frame (row-major 3x4 Twc, the layout of the reference's 0011.txt, Tracking.cc:449-479) and
Segmentation/%06d.png 16-bit instance ids 1000+inst (src/Frame.cc:1023-1043).

The scene is geometrically exact: a static surface whose depth depends on the image row only, z(y) linear 60 m (top) ..
6 m (bottom), seen by a camera that translates along +x; a translation t moves the row y by fx*t/z(y) pixels and the
right camera sees it a further bf/z(y) to the left.  Every image is ONE bilinear resampling of a wide base texture, so
errors do not accumulate over frames.  Two textured boxes at constant depth move with their own velocity (dynamic
objects: outliers for the static tracker, the objects of ObjectTracking.txt).
"""
import os

import numpy as np

from .synth import Rng, _value_noise, KITTI_K, KITTI_BF


def _texture(rng, w, h, n_rect):
    img = np.full((h, w), 128.0)
    for cell, amp in ((32, 64.0), (16, 32.0), (8, 16.0), (3, 14.0)):
        img += _value_noise(rng, w, h, cell, amp)
    rw = rng.integers(n_rect, 8, 65); rh = rng.integers(n_rect, 8, 65)
    rx = rng.integers(n_rect, 0, w); ry = rng.integers(n_rect, 0, h)
    rg = rng.integers(n_rect, 0, 256)
    for i in range(n_rect):
        img[ry[i]:ry[i] + rh[i], rx[i]:rx[i] + rw[i]] = rg[i]
    return np.clip(np.rint(img), 0, 255)


def _resample_rows(base, shift, w):
    """out[y, x] = base[y, x + shift[y]] (bilinear), for x in [0, w)"""
    h = base.shape[0]
    xs = np.arange(w)[None, :] + shift[:, None]
    x0 = np.floor(xs).astype(np.int64)
    fx = xs - x0
    x0 = np.clip(x0, 0, base.shape[1] - 2)
    rows = np.arange(h)[:, None]
    return base[rows, x0] * (1 - fx) + base[rows, x0 + 1] * fx


def kitti_texture():
    """The one real KITTI frame in the repository (tests/golden/kitti_000212_gray.png, the grey-converted data file that ships
    in the reference's root) as a float array, for generate(texture=...)."""
    from PIL import Image
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kitti_000212_gray.png")
    return np.asarray(Image.open(p)).astype(np.float64)


def _texture_from_image(img, w, h, shift):
    """A base texture of w x h from a real image: rows cropped / mirror-padded to h, columns continued by mirroring (a
    reflection keeps the local statistics - corners per pixel, contrast - of the photograph); `shift` moves the start column so
    that sequences with different seeds see different parts first."""
    img = np.asarray(img, np.float64)
    if img.shape[0] < h:
        img = np.pad(img, ((0, h - img.shape[0]), (0, 0)), mode="reflect")
    img = img[:h]
    period = np.concatenate([img, img[:, ::-1]], 1)
    reps = (w + shift) // period.shape[1] + 2
    return np.tile(period, (1, reps))[:, shift:shift + w].copy()


def generate(n_frames=20, seed=4, w=1242, h=375, step=0.08, n_boxes=2, K=KITTI_K, bf=KITTI_BF, texture=None):
    """Returns a dict: left/right uint8 [n, h, w], twc float64 [n, 3, 4] (ground truth), boxes (per frame, per box:
    x1 y1 x2 y2 in the left image, depth), seg uint16 [n, h, w], K, bf.  texture: None = the seeded value-noise + rectangles
    texture; an image array = that photograph as the static surface's texture (same exact geometry)."""
    rng = Rng(0x51070000 + seed)
    fx, fy, cx, cy = [float(v) for v in K]
    z = 60.0 + (6.0 - 60.0) * (np.arange(h) / (h - 1))
    # camera x positions: nominal step with a seeded +-25 % variation so that the constant-velocity prediction is never exact
    steps = step * (1.0 + rng.uniform(n_frames, -0.25, 0.25))
    steps[0] = 0.0
    tx = np.cumsum(steps)
    margin = int(np.ceil(fx * tx[-1] / z.min() + bf / z.min())) + 8
    if texture is None:
        base = _texture(rng, w + margin, h, int(400 * (w + margin) / 1242))
    else:
        base = _texture_from_image(texture, w + margin, h, int(rng.integers(1, 0, 997)[0]))
    box_tex, box_geo = [], []
    for b in range(n_boxes):
        bw, bh = int(rng.integers(1, 90, 140)[0]), int(rng.integers(1, 50, 80)[0])
        zb = float(rng.uniform(1, 9.0, 16.0)[0])
        v0 = int(rng.integers(1, h // 3, h - bh - 20)[0])
        x0 = float(rng.uniform(1, -3.0, 3.0)[0])           # metres, world
        vel = float(rng.uniform(1, 0.05, 0.25)[0]) * (1 if b % 2 == 0 else -1)
        box_tex.append(_texture(rng, bw, bh, 12))
        box_geo.append((bw, bh, zb, v0, x0, vel))
    left = np.zeros((n_frames, h, w), np.uint8); right = np.zeros_like(left)
    seg = np.zeros((n_frames, h, w), np.uint16)
    twc = np.zeros((n_frames, 3, 4))
    boxes = []
    for k in range(n_frames):
        sh = fx * tx[k] / z
        L = _resample_rows(base, sh, w)
        R = _resample_rows(base, sh + bf / z, w)
        fb = []
        for b, (bw, bh, zb, v0, x0, vel) in enumerate(box_geo):
            u = int(round(fx * (x0 + vel * k - tx[k]) / zb + cx))
            d = int(round(bf / zb))
            for img, uu, is_left in ((L, u, True), (R, u - d, False)):
                a0, a1 = max(uu, 0), min(uu + bw, w)
                if a1 > a0:
                    img[v0:v0 + bh, a0:a1] = box_tex[b][:, a0 - uu:a1 - uu]
                    if is_left:
                        seg[k, v0:v0 + bh, a0:a1] = 1000 + b
            fb.append((u, v0, u + bw, v0 + bh, zb, x0 + vel * k))
        boxes.append(fb)
        left[k] = np.clip(np.rint(L), 0, 255).astype(np.uint8)
        right[k] = np.clip(np.rint(R), 0, 255).astype(np.uint8)
        twc[k, :3, :3] = np.eye(3); twc[k, 0, 3] = tx[k]
    return {"left": left, "right": right, "twc": twc, "boxes": boxes, "seg": seg, "K": (fx, fy, cx, cy), "bf": float(bf)}


# ---------------------------------------------------------------------------------------------------------------------
# A KITTI-like drive: the camera moves FORWARD with a slowly changing yaw through a corridor of exactly ray-cast planes - a ground
# plane 1.65 m below the camera, two side walls, a ceiling far above, an end wall ahead - textured from a photograph; the right
# camera sees the same world from one baseline to the right, so depth and disparity are analytic per pixel.  Forward motion makes
# SearchByProjection(cur, last) take its bForward branch (ORBmatcher.cc:1634-1675), keypoints change octave between frames and
# local-map points leave the frustum - what the lateral scene of generate() never exercises.  Moving objects are textured
# rectangles (the front faces of cuboids) with their own world velocity, rendered by the same ray casting in both cameras.
# ---------------------------------------------------------------------------------------------------------------------
def _mip_chain(tex, levels=5):
    out = [np.asarray(tex, np.float64)]
    for _ in range(levels - 1):
        t = out[-1]
        h, w = t.shape[0] // 2 * 2, t.shape[1] // 2 * 2
        out.append(0.25 * (t[0:h:2, 0:w:2] + t[1:h:2, 0:w:2] + t[0:h:2, 1:w:2] + t[1:h:2, 1:w:2]))
    return out


def _sample(mips, u, v, footprint):
    """bilinear lookup of the (mirror-tiled) texture at texel coordinates (u, v), mip level by the pixel footprint in texels"""
    lvl = np.clip(np.floor(np.log2(np.maximum(footprint, 1.0)) + 0.5), 0, len(mips) - 1).astype(np.int64)
    out = np.zeros(u.shape)
    for l, t in enumerate(mips):
        m = lvl == l
        if not m.any():
            continue
        th, tw = t.shape
        uu, vv = u[m] / (1 << l) - 0.5, v[m] / (1 << l) - 0.5
        x0 = np.floor(uu).astype(np.int64); y0 = np.floor(vv).astype(np.int64)
        fx, fy = uu - x0, vv - y0

        def wrap(i, n):            # mirror tiling: ... 2 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...
            i = np.mod(i, 2 * n)
            return np.where(i < n, i, 2 * n - 1 - i)
        xa, xb, ya, yb = wrap(x0, tw), wrap(x0 + 1, tw), wrap(y0, th), wrap(y0 + 1, th)
        out[m] = (t[ya, xa] * (1 - fx) + t[ya, xb] * fx) * (1 - fy) + (t[yb, xa] * (1 - fx) + t[yb, xb] * fx) * fy
    return out


def _render(cam_pos, yaw, K, w, h, mips, objs, texel_per_m=40.0):
    """One pinhole view from cam_pos (world, y down) rotated by `yaw` about y.  Returns (image float, depth along the camera's z,
    object id per pixel (-1: static world)).  objs: list of (centre x, bottom y, z of the face, width, height, texture mips)."""
    fx, fy, cx, cy = K
    us, vs = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(us - cx) / fx, (vs - cy) / fy, np.ones_like(us)], -1)           # camera-frame ray, z = 1
    c, s_ = np.cos(yaw), np.sin(yaw)
    dw = np.stack([c * dc[..., 0] + s_ * dc[..., 2], dc[..., 1], -s_ * dc[..., 0] + c * dc[..., 2]], -1)   # Rwc = Ry(yaw)
    best_t = np.full((h, w), np.inf)
    img = np.zeros((h, w)); oid = np.full((h, w), -1, np.int64)
    half_w, ground, ceil_y, z_end = 9.0, 1.65, -7.0, 400.0

    def hit(t, ucoord, vcoord, valid, tex, scale, ident):
        nonlocal best_t, img, oid
        ok = valid & (t > 0.05) & (t < best_t)
        if not ok.any():
            return
        foot = (t[ok] / fx) * scale                                              # texels per image pixel (isotropic estimate)
        img[ok] = _sample(tex, ucoord[ok] * scale, vcoord[ok] * scale, foot)
        best_t = np.where(ok, t, best_t)
        oid[ok] = ident
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground y = ground (texture over x, z), ceiling, walls x = +-half_w (texture over z, y), end wall
        t = (ground - cam_pos[1]) / dw[..., 1]
        hit(t, cam_pos[0] + t * dw[..., 0] + 1000.0, cam_pos[2] + t * dw[..., 2], dw[..., 1] > 1e-9, mips, texel_per_m, -1)
        t = (ceil_y - cam_pos[1]) / dw[..., 1]
        hit(t, cam_pos[0] + t * dw[..., 0] + 1000.0, cam_pos[2] + t * dw[..., 2] + 517.0, dw[..., 1] < -1e-9, mips, texel_per_m * 0.5, -1)
        for sign, off in ((1.0, 211.0), (-1.0, 733.0)):
            t = (sign * half_w - cam_pos[0]) / dw[..., 0]
            hit(t, cam_pos[2] + t * dw[..., 2] + off, cam_pos[1] + t * dw[..., 1] + 50.0, sign * dw[..., 0] > 1e-9, mips, texel_per_m, -1)
        t = (z_end - cam_pos[2]) / dw[..., 2]
        hit(t, cam_pos[0] + t * dw[..., 0] + 300.0, cam_pos[1] + t * dw[..., 1] + 80.0, dw[..., 2] > 1e-9, mips, texel_per_m * 0.25, -1)
        for k, (ox, oy, oz, ow, oh, otex) in enumerate(objs):
            t = (oz - cam_pos[2]) / dw[..., 2]
            px, py = cam_pos[0] + t * dw[..., 0], cam_pos[1] + t * dw[..., 1]
            inside = (dw[..., 2] > 1e-9) & (px >= ox - ow / 2) & (px < ox + ow / 2) & (py >= oy - oh) & (py < oy)
            hit(t, px - (ox - ow / 2), py - (oy - oh), inside, otex, 60.0, k)
    depth = best_t * 1.0                                                          # camera z of the hit: t * dc.z with dc.z = 1
    return img, depth, oid


def generate_drive(n_frames=20, seed=4, w=1242, h=375, speed=0.7, n_objects=2, K=KITTI_K, bf=KITTI_BF, texture=None, yaw_rate_deg=0.6):
    """A forward drive with yaw: dict like generate() - left / right uint8 [n, h, w], twc [n, 3, 4] (Twc rows, ground truth), seg
    uint16 [n, h, w] (MOTS ids 1000 + object), boxes per frame, K, bf - plus `labels` (per frame the KITTI label fields)."""
    rng = Rng(0x51071000 + seed)
    fx, fy, cx, cy = [float(v) for v in K]
    b = float(bf) / fx
    if texture is None:
        texture = _texture(rng, 1536, 768, 600)
    mips = _mip_chain(texture)
    steps = speed * (1.0 + rng.uniform(n_frames, -0.25, 0.25)); steps[0] = 0.0
    # the vehicle pulls away: the frame after the initialisation is tracked from an identity velocity (no ORBvoc.bin, tracker.py), which
    # only reaches a short first step
    if n_frames > 1:
        steps[1] *= 0.3
    if n_frames > 2:
        steps[2] *= 0.65
    # yaw: a slow sine plus a seeded phase, so that the constant-velocity model is never exact
    ph = float(rng.uniform(1, 0.0, 6.28)[0])
    yaw = np.deg2rad(yaw_rate_deg) * 8.0 * (np.sin(np.arange(n_frames) / 8.0 + ph) - np.sin(ph))
    pos = np.zeros((n_frames, 3))
    for k in range(1, n_frames):
        pos[k] = pos[k - 1] + steps[k] * np.array([np.sin(yaw[k - 1]), 0.0, np.cos(yaw[k - 1])])
    # objects: a vehicle ahead driving on at about the camera's speed, and one standing at the side that the camera passes
    objs0 = []
    for o in range(n_objects):
        ow, oh = float(rng.uniform(1, 1.6, 2.2)[0]), float(rng.uniform(1, 1.3, 1.7)[0])
        if o % 2 == 0:
            ox, oz, vz = float(rng.uniform(1, -1.5, 1.5)[0]), float(rng.uniform(1, 11.0, 15.0)[0]), speed * float(rng.uniform(1, 0.85, 1.1)[0])
        else:
            ox, oz, vz = float(rng.uniform(1, 3.0, 5.0)[0]) * (1 if o % 4 == 1 else -1), float(rng.uniform(1, 22.0, 30.0)[0]), 0.0
        otex = _mip_chain(_texture(rng, int(ow * 60) + 2, int(oh * 60) + 2, 14), 3)
        objs0.append((ox, oz, vz, ow, oh, otex))
    left = np.zeros((n_frames, h, w), np.uint8); right = np.zeros_like(left)
    seg = np.zeros((n_frames, h, w), np.uint16)
    twc = np.zeros((n_frames, 3, 4))
    boxes, labels = [], []
    for k in range(n_frames):
        objs = [(ox, 1.65, oz + vz * k, ow, oh, otex) for (ox, oz, vz, ow, oh, otex) in objs0]
        c, s_ = np.cos(yaw[k]), np.sin(yaw[k])
        Rwc = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
        L, _, oid = _render(pos[k], yaw[k], (fx, fy, cx, cy), w, h, mips, objs)
        R, _, _ = _render(pos[k] + Rwc @ np.array([b, 0.0, 0.0]), yaw[k], (fx, fy, cx, cy), w, h, mips, objs)
        left[k] = np.clip(np.rint(L), 0, 255).astype(np.uint8); right[k] = np.clip(np.rint(R), 0, 255).astype(np.uint8)
        seg[k][oid >= 0] = (1000 + oid[oid >= 0]).astype(np.uint16)
        twc[k, :3, :3] = Rwc; twc[k, :, 3] = pos[k]
        fb, fl = [], []
        for o, (ox, oy, oz, ow, oh, _) in enumerate(objs):
            ys, xs = np.nonzero(oid == o)
            if len(xs) < 200:
                fb.append(None)
                continue
            x1, x2, y1, y2 = float(xs.min()), float(xs.max() + 1), float(ys.min()), float(ys.max() + 1)
            # the cuboid: front face = the rendered rectangle, BOX_DEPTH_M deep; bottom centre and rotation_y in the camera frame
            cw = Rwc.T @ (np.array([ox, oy, oz + 0.5 * BOX_DEPTH_M]) - pos[k])
            fb.append((x1, y1, x2, y2, float(cw[2]), float(cw[0])))
            fl.append((o, x1, y1, x2, y2, oh, BOX_DEPTH_M, ow, float(cw[0]), float(cw[1]), float(cw[2]), float(-yaw[k])))
        boxes.append(fb); labels.append(fl)
    return {"left": left, "right": right, "twc": twc, "boxes": boxes, "labels": labels, "seg": seg, "K": (fx, fy, cx, cy), "bf": float(bf)}


BOX_DEPTH_M = 0.5      # extent of a generated box along the viewing direction (the label's `w` at rotation_y = 0)


def frame_labels(seq, k):
    """The KITTI-tracking label fields of frame k's boxes: (track, x1, y1, x2, y2, h, w, l, X, Y, Z, ry).  A box is a textured
    plane at depth zb facing the camera; its cuboid has the plane as front face: l (along x at ry = 0) = the plane's width,
    h its height, w = BOX_DEPTH_M, bottom centre (X, Y, Z = zb + w / 2)."""
    if "labels" in seq:
        return list(seq["labels"][k])
    fx, fy, cx, cy = seq["K"]
    out = []
    for b, (x1, y1, x2, y2, zb, xw) in enumerate(seq["boxes"][k]):
        if x2 <= 0 or x1 >= seq["left"].shape[2]:
            continue
        hm, lm = (y2 - y1) * zb / fy, (x2 - x1) * zb / fx
        X = (0.5 * (x1 + x2) - cx) * zb / fx; Y = (y2 - cy) * zb / fy
        out.append((b, float(x1), float(y1), float(x2), float(y2), hm, BOX_DEPTH_M, lm, X, Y, zb + 0.5 * BOX_DEPTH_M, 0.0))
    return out


def frame_detections(seq, k):
    """Frame k's offline detections as Frame::OfflineDetectObject hands them to the tracker (SLOT.MODE 4)."""
    from .object_tracker import detection_from_label
    return [detection_from_label(*lab) for lab in frame_labels(seq, k)]


def frame_mask(seq, k):
    """Frame::ReadKittiSegmentationImage (src/Frame.cc:1004-1043) on frame k's MOTS ids: 0 background, 255 for 10000 (ignored),
    instance + 1 for ids 1000..1999."""
    seg = seq["seg"][k]
    m = np.zeros(seg.shape, np.uint8)
    m[seg == 10000] = 255
    car = (seg >= 1000) & (seg < 2000)
    m[car] = (seg[car] % 1000 + 1).astype(np.uint8)
    return m


def write(seq_dir, seq, dt=0.1, pgm=False):
    """Writes `seq` (from generate()) in the reference's on-disk layout.  pgm=True adds binary PGM copies of the stereo images
    next to the PNGs (examples/stereo_kitti.cpp reads those: the build image has no PNG decoder for C++)."""
    from PIL import Image
    for d in ("image_02", "image_03", "Segmentation"):
        os.makedirs(os.path.join(seq_dir, d), exist_ok=True)
    n = len(seq["left"])
    for k in range(n):
        Image.fromarray(seq["left"][k]).save(os.path.join(seq_dir, "image_02", "%06d.png" % k))
        Image.fromarray(seq["right"][k]).save(os.path.join(seq_dir, "image_03", "%06d.png" % k))
        Image.fromarray(seq["seg"][k]).save(os.path.join(seq_dir, "Segmentation", "%06d.png" % k))
        if pgm:
            for d, im in (("image_02", seq["left"][k]), ("image_03", seq["right"][k])):
                with open(os.path.join(seq_dir, d, "%06d.pgm" % k), "wb") as f:
                    f.write(b"P5\n%d %d\n255\n" % (im.shape[1], im.shape[0]))
                    f.write(np.ascontiguousarray(im).tobytes())
    with open(os.path.join(seq_dir, "timestamp.txt"), "w") as f:
        for k in range(n):
            f.write("%.6f\n" % (k * dt))
    with open(os.path.join(seq_dir, "poses.txt"), "w") as f:
        for k in range(n):
            f.write(" ".join("%.9g" % v for v in seq["twc"][k].reshape(12)) + "\n")
    fx, fy, cx, cy = seq["K"]
    with open(os.path.join(seq_dir, "ObjectTracking.txt"), "w") as f:
        # frame track type trunc occl alpha x1 y1 x2 y2 h w l X Y Z ry   (camera-frame X Y Z of the box bottom centre)
        for k in range(n):
            for lab in frame_labels(seq, k):
                f.write("%d %d Car 0 0 0 %.2f %.2f %.2f %.2f %.3f %.3f %.3f %.3f %.3f %.3f %.3f\n" % ((k,) + lab))
    with open(os.path.join(seq_dir, "calib.txt"), "w") as f:
        f.write("Camera.fx: %.9g\nCamera.fy: %.9g\nCamera.cx: %.9g\nCamera.cy: %.9g\nCamera.bf: %.9g\nThDepth: 35\n" % (fx, fy, cx, cy, seq["bf"]))


def write_pgm(seq_dir, seq, dt=0.1):
    """Only what examples/stereo_kitti*.cpp read: binary PGM stereo images, timestamp.txt, calib.txt (no PNG encoding)."""
    for d in ("image_02", "image_03"):
        os.makedirs(os.path.join(seq_dir, d), exist_ok=True)
    n = len(seq["left"])
    for k in range(n):
        for d, im in (("image_02", seq["left"][k]), ("image_03", seq["right"][k])):
            with open(os.path.join(seq_dir, d, "%06d.pgm" % k), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (im.shape[1], im.shape[0]))
                f.write(np.ascontiguousarray(im).tobytes())
    with open(os.path.join(seq_dir, "timestamp.txt"), "w") as f:
        for k in range(n):
            f.write("%.6f\n" % (k * dt))
    fx, fy, cx, cy = seq["K"]
    with open(os.path.join(seq_dir, "calib.txt"), "w") as f:
        f.write("Camera.fx: %.9g\nCamera.fy: %.9g\nCamera.cx: %.9g\nCamera.cy: %.9g\nCamera.bf: %.9g\nThDepth: 35\n" % (fx, fy, cx, cy, seq["bf"]))


def _to_gray(a, rgb_order=True):
    """cv::cvtColor RGB2GRAY / BGR2GRAY of OpenCV 3.4 on 8-bit data: (R*4899 + G*9617 + B*1868 + 8192) >> 14
    (Tracking.cc:1016-1037; `Camera.RGB` picks which channel gets the R weight)."""
    a = a.astype(np.int64)
    r, g, b = (a[..., 0], a[..., 1], a[..., 2]) if rgb_order else (a[..., 2], a[..., 1], a[..., 0])
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def load(seq_dir, max_frames=None):
    """LoadImages of stereo_kitti.cc:170-230: timestamps decide the frame count; images are %06d.png in image_02 / image_03."""
    from PIL import Image
    with open(os.path.join(seq_dir, "timestamp.txt")) as f:
        stamps = [float(s.split()[0]) for s in f if s.strip()]
    if max_frames is not None:
        stamps = stamps[:max_frames]
    left, right = [], []
    for k in range(len(stamps)):
        for lst, d in ((left, "image_02"), (right, "image_03")):
            a = np.asarray(Image.open(os.path.join(seq_dir, d, "%06d.png" % k)))
            lst.append(_to_gray(a) if a.ndim == 3 else a.astype(np.uint8))
    calib = {}
    p = os.path.join(seq_dir, "calib.txt")
    if os.path.exists(p):
        with open(p) as f:
            for line in f:
                k, v = line.split(":")
                calib[k.strip()] = float(v)
    return {"left": left, "right": right, "timestamps": stamps, "calib": calib}
