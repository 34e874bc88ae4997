"""Image assembly after the renderer: scatter per-ray colours
back into H x W by `ray_mask`, background fill, 8-bit quantisation.

`unpack_to_image` / `to_8b_image` reproduce run.py:46-63 and image_util.py:19-20 byte for byte
on the host; `assemble_uint8_device` does the same on the GPU so that only uint8 pixels cross
PCIe (3 B/pixel instead of 16 B/ray)."""
import os
import shutil

import numpy as np
import torch


def to_8b_image(image):
    return (255. * np.clip(image, 0., 1.)).astype(np.uint8)


def to_8b3ch_image(image):
    im = to_8b_image(image)
    return np.stack([im, im, im], axis=-1) if im.ndim == 2 else np.concatenate([im] * 3, axis=-1)


def unpack_to_image(width, height, ray_mask, bgcolor, rgb, alpha, truth=None):
    rgb_image = np.full((height * width, 3), bgcolor, dtype='float32')
    rgb_image[ray_mask] = rgb
    rgb_image = to_8b_image(rgb_image.reshape((height, width, 3)))
    truth_image = np.full((height * width, 3), bgcolor, dtype='float32')
    if truth is not None:
        truth_image[ray_mask] = truth
        truth_image = to_8b_image(truth_image.reshape((height, width, 3)))
    alpha_map = np.zeros((height * width), dtype='float32')
    alpha_map[ray_mask] = alpha
    return rgb_image, to_8b3ch_image(alpha_map.reshape((height, width))), truth_image


def assemble_uint8_device(width, height, ray_index, bgcolor, rgb, alpha):
    """Same result as unpack_to_image, computed on the device.  ray_index: int64 [R] flat pixel
    index of every ray (nonzero(ray_mask)); rgb [R,3], alpha [R] on the GPU.  -> uint8 tensors
    (rgb image [H,W,3], alpha image [H,W,3]) still on the GPU."""
    dev = rgb.device
    bg = torch.as_tensor(np.asarray(bgcolor, dtype=np.float32), device=dev)
    img = bg.expand(height * width, 3).clone()
    img[ray_index] = rgb
    a = torch.zeros(height * width, device=dev, dtype=torch.float32)
    a[ray_index] = alpha
    # (255. * clip(x, 0, 1)).astype(uint8): multiply in fp32, truncate toward zero
    q = (img.clamp(0., 1.) * 255.).to(torch.uint8).view(height, width, 3)
    qa = (a.clamp(0., 1.) * 255.).to(torch.uint8).view(height, width, 1).expand(height, width, 3)
    return q, qa.contiguous()


class ImageWriter:
    """PNG dump with the reference's folder layout (image_util.py:53-75):
    <output_dir>/<exp_name>/NNNNNN.png, directory recreated on start."""

    def __init__(self, output_dir, exp_name):
        self.image_dir = os.path.join(output_dir, exp_name)
        print('The rendering is saved in ' + self.image_dir)
        if os.path.exists(self.image_dir):
            shutil.rmtree(self.image_dir)
        os.makedirs(self.image_dir, exist_ok=True)
        self.frame_idx = -1

    def append(self, image, img_name=None):
        from PIL import Image
        self.frame_idx += 1
        if img_name is None:
            img_name = f'{self.frame_idx:06d}'
        Image.fromarray(image).save(f'{self.image_dir}/{img_name}.png')
        return self.frame_idx, img_name

    def finalize(self):
        pass
