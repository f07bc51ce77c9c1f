"""Scene files -> numpy (reference dataloading/common.py:8-314): frames, LLFF pose blocks, monocular / sensor depth maps, and
the pose normalisations applied before training (recentre, optional spherify).  PIL decodes the images; no imageio / cv2."""
import os

import numpy as np
from PIL import Image

_IMAGE_SUFFIXES = ('JPG', 'jpg', 'png')


def _frames_in(folder):
    return [f for f in sorted(os.listdir(folder)) if f.endswith(_IMAGE_SUFFIXES)]


def _read_rgb(path):
    """8-bit file -> (h, w, 3|4) uint8 array, no gamma handling (the reference reads PNGs with ignoregamma=True)."""
    with Image.open(path) as im:
        if im.mode not in ('RGB', 'RGBA'):
            im = im.convert('RGB')
        return np.asarray(im)


def _minify(basedir, factors=[], resolutions=[], img_folder='images'):
    """Make sure the down-sized copies images_F / images_WxH exist.  The reference shells out to ImageMagick `mogrify -resize`
    (common.py:8-57); here PIL resamples (Lanczos, ImageMagick's default when shrinking), so a folder created by this function
    is close to, not bit-identical with, one created by the reference.  Existing folders are used as they are."""
    src = os.path.join(basedir, img_folder)
    for r in list(factors) + list(resolutions):
        name = img_folder + ('_{}'.format(r) if isinstance(r, int) else '_{}x{}'.format(r[1], r[0]))
        dst = os.path.join(basedir, name)
        if os.path.exists(dst):
            continue
        print('Minifying', r, basedir)
        os.makedirs(dst)
        for f in _frames_in(src):
            with Image.open(os.path.join(src, f)) as im:
                size = (round(im.width / r), round(im.height / r)) if isinstance(r, int) else (r[1], r[0])
                im.resize(size, Image.LANCZOS).save(os.path.join(dst, os.path.splitext(f)[0] + '.png'))


def _crop_borders(basedir, crop_size):
    """images/ -> images_cropped/: cut `crop_size` rows (and the aspect-matching number of columns) off every side, resize back
    (common.py:67-88).  -> (crop_ratio, focal_crop_factor)."""
    dst = os.path.join(basedir, 'images_cropped')
    os.makedirs(dst, exist_ok=True)
    H = None
    for f in _frames_in(os.path.join(basedir, 'images')):
        img = _read_rgb(os.path.join(basedir, 'images', f))
        H, W = img.shape[:2]
        cw = int(crop_size * W / H)
        Image.fromarray(img[crop_size:H - crop_size, cw:W - cw]).resize((W, H)).save(os.path.join(dst, f))
    print('=======images cropped=======')
    return crop_size / H, (H - 2 * crop_size) / H


def _load_data(basedir, factor=None, width=None, height=None, load_imgs=True, crop_size=0, load_colmap_poses=True):
    """-> poses (3,5,n) [rotation|centre|(h,w,focal)] in LLFF axes, bds (2,n), imgs (h,w,3,n) float64 in [0,1], file names,
    crop_ratio, focal_crop_factor (common.py:59-141).  With load_imgs=False: (poses, bds)."""
    poses = bds = None
    if load_colmap_poses:
        arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
        poses = arr[:, :-2].reshape(-1, 3, 5).transpose(1, 2, 0)
        bds = arr[:, -2:].transpose(1, 0)
    img_folder, crop_ratio, focal_crop_factor = 'images', 1, 1
    if crop_size != 0:
        img_folder = 'images_cropped'
        crop_ratio, focal_crop_factor = _crop_borders(basedir, crop_size)

    first = _frames_in(os.path.join(basedir, img_folder))[0]
    full = _read_rgb(os.path.join(basedir, img_folder, first)).shape
    sfx = ''
    if factor is not None:
        sfx = '_{}'.format(factor)
        _minify(basedir, factors=[factor], img_folder=img_folder)
    elif height is not None:
        factor = full[0] / float(height)
        width = int(full[1] / factor)
        _minify(basedir, resolutions=[[height, width]], img_folder=img_folder)
        sfx = '_{}x{}'.format(width, height)
    elif width is not None:
        factor = full[1] / float(width)
        height = int(full[0] / factor)
        _minify(basedir, resolutions=[[height, width]], img_folder=img_folder)
        sfx = '_{}x{}'.format(width, height)
    else:
        factor = 1

    imgdir = os.path.join(basedir, img_folder + sfx)
    if not os.path.exists(imgdir):
        print(imgdir, 'does not exist, returning')
        return
    names = _frames_in(imgdir)
    shape = _read_rgb(os.path.join(imgdir, names[0])).shape
    if load_colmap_poses:
        if poses.shape[-1] != len(names):
            print('Mismatch between imgs {} and poses {} !!!!'.format(len(names), poses.shape[-1]))
            return
        poses[:2, 4, :] = np.array(shape[:2]).reshape(2, 1)      # (h, w) of the frames actually loaded
        poses[2, 4, :] = poses[2, 4, :] * 1. / factor            # focal follows the down-sizing
    if not load_imgs:
        return poses, bds
    imgs = np.stack([_read_rgb(os.path.join(imgdir, f))[..., :3] / 255. for f in names], -1)
    print('Loaded image data', imgs.shape, *([poses[:, -1, 0]] if load_colmap_poses else []))
    return poses, bds, imgs, names, crop_ratio, focal_crop_factor


# ---------------------------------------------------------------------------------------------- pose normalisation
def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """3x4 [x|y|z|pos] with z along `z`, x = up x z, y = z x x."""
    z = normalize(z)
    x = normalize(np.cross(up, z))
    return np.stack([x, normalize(np.cross(z, x)), z, pos], 1)


def poses_avg(poses):
    """The 'average' camera of (n,3,5) poses: mean centre, summed viewing / up axes (common.py:155-165)."""
    frame = viewmatrix(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], 1)


def _to44(p34):
    last = np.broadcast_to(np.array([0, 0, 0, 1.]), p34.shape[:-2] + (1, 4))
    return np.concatenate([p34, last], -2)


def recenter_poses(poses):
    """Express all (n,3,5) poses in the average camera's frame (common.py:142-154)."""
    out = poses + 0
    out[:, :3, :4] = (np.linalg.inv(_to44(poses_avg(poses)[:3, :4])) @ _to44(poses[:, :3, :4]))[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """Recentre on the point closest to all optical axes, scale the cameras onto the unit sphere and build the 120-view circular
    render path (common.py:175-232).  -> (poses (n,3,5), render path (120,3,5), bds); `bds` is scaled in place."""
    axis, centre = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - axis * np.transpose(axis, [0, 2, 1])
    b = -A @ centre
    focus = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    up = normalize((poses[:, :3, 3] - focus).mean(0))
    e1 = normalize(np.cross([.1, .2, .3], up))
    e2 = normalize(np.cross(up, e1))
    frame = np.stack([e1, e2, up, focus], 1)
    local = np.linalg.inv(_to44(frame[None])) @ _to44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(local[:, :3, 3]), -1)))
    sc = 1. / rad
    local[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(local[:, :3, 3], 0)[2]
    ring = np.sqrt(rad ** 2 - zh ** 2)
    path = []
    for th in np.linspace(0., 2. * np.pi, 120):
        eye = np.array([ring * np.cos(th), ring * np.sin(th), zh])
        z = normalize(eye)
        x = normalize(np.cross(z, np.array([0, 0, -1.])))
        path.append(np.stack([x, normalize(np.cross(z, x)), z, eye], 1))
    path = np.stack(path, 0)
    hwf = poses[0, :3, -1:]
    path = np.concatenate([path, np.broadcast_to(hwf, path[:, :3, -1:].shape)], -1)
    local = np.concatenate([local[:, :3, :4], np.broadcast_to(hwf, local[:, :3, -1:].shape)], -1)
    return local, path, bds


# ---------------------------------------------------------------------------------------------- depth maps
def _frame_id(image_name):
    return image_name.split('.')[0]


def _resize_bilinear(depth, H, W):
    """cv2.resize(depth, (W, H)) default (bilinear, half-pixel centres, no antialias)."""
    import torch
    import torch.nn.functional as F
    t = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32))[None, None]
    return F.interpolate(t, size=(H, W), mode='bilinear', align_corners=False)[0, 0].numpy()


def _resize_nearest(depth, H, W):
    ys = np.minimum((np.arange(H) * (depth.shape[0] / H)).astype(np.int64), depth.shape[0] - 1)
    xs = np.minimum((np.arange(W) * (depth.shape[1] / W)).astype(np.int64), depth.shape[1] - 1)
    return depth[ys][:, xs]


def load_gt_depths(image_list, datadir, H=None, W=None, crop_ratio=1):
    """Sensor depth: depth/<frame>.png, 16-bit millimetres -> metres float32 (common.py:235-258)."""
    out = []
    for name in image_list:
        with Image.open(os.path.join(datadir, 'depth', '{}.png'.format(_frame_id(name)))) as im:
            depth = np.asarray(im).astype(np.float32) / 1000
        if crop_ratio != 1:
            h, w = depth.shape
            ch, cw = int(h * crop_ratio), int(w * crop_ratio)
            depth = depth[ch:h - ch, cw:w - cw]
        out.append(depth if H is None else _resize_nearest(depth, H, W))
    return np.stack(out)


def load_depths(image_list, datadir, H=None, W=None):
    """<frame>_depth.npy or depth_<frame>.npy (common.py:259-275)."""
    out = []
    for name in image_list:
        path = os.path.join(datadir, '{}_depth.npy'.format(_frame_id(name)))
        if not os.path.exists(path):
            path = os.path.join(datadir, 'depth_{}.npy'.format(_frame_id(name)))
        depth = np.load(path)
        out.append(depth if H is None else _resize_bilinear(depth, H, W))
    return np.stack(out)


def load_images(image_list, datadir):
    return np.stack([np.load(os.path.join(datadir, '{}.npy'.format(_frame_id(name)))) for name in image_list])


def load_depths_npz(image_list, datadir, H=None, W=None, norm=False):
    """Monocular depth: depth_<frame>.npz, key 'pred', (1,h,w) or (h,w) (common.py:286-314).  norm=True maps every frame's
    median / mean-absolute-deviation onto those of the whole stack."""
    out = []
    for name in image_list:
        depth = np.load(os.path.join(datadir, 'depth_{}.npz'.format(_frame_id(name))))['pred']
        if depth.shape[0] == 1:
            depth = depth[0]
        out.append(depth if H is None else _resize_bilinear(depth, H, W))
    depths = np.stack(out)
    if norm:
        t_all = np.median(depths)
        s_all = np.mean(np.abs(depths - t_all))
        med = [np.median(d) for d in depths]
        depths = np.stack([s_all * (d - t) / np.mean(np.abs(d - t)) + t_all for d, t in zip(depths, med)])
    return depths
