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


def assemble_uint8_device(width, height, ray_index, bgcolor, rgb, alpha, want_alpha=True):
    """Same result as unpack_to_image, computed by one HIP kernel (csrc/image.hip).  ray_index: int64 [R] ascending
    flat pixel index of every ray (nonzero(ray_mask)); rgb [R,3], alpha [R] on the GPU; bgcolor: cfg.bgcolor / 255.
    -> uint8 tensors (rgb image [H,W,3], alpha image [H,W,3] or None) still on the GPU."""
    from . import _lib, ops
    dev = rgb.device
    if dev.type != 'cuda':
        raise RuntimeError('assemble_uint8_device: needs GPU tensors (unpack_to_image is the host function)')
    R = int(ray_index.numel())
    out = torch.empty(height, width, 3, device=dev, dtype=torch.uint8)
    out_a = torch.empty(height, width, 3, device=dev, dtype=torch.uint8) if want_alpha else None
    _bg, pbg = ops._host_f32(np.asarray(bgcolor, dtype=np.float32), 3)
    with ops._guard_dev(dev):
        rc = _lib.lib().occnerf_assemble_image(
            ops._chk(rgb.contiguous(), torch.float32, 'rgb') if R else None,
            (ops._chk(alpha.contiguous(), torch.float32, 'alpha') if R else None) if want_alpha else None,
            ops._chk(ray_index, torch.int64, 'ray_index') if R else None, R, int(height), int(width), pbg,
            out.data_ptr(), None if out_a is None else out_a.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, 'assemble_image')
    return out, out_a


class ImageWriter:
    """PNG dump with the reference's folder layout (image_util.py:53-75): <output_dir>/<exp_name>/NNNNNN.png,
    directory recreated on start.  `append` takes a host array like the reference's; `append_device` takes a uint8
    GPU image: it is copied into one of a few pinned staging buffers without blocking the caller, and a writer thread
    waits for the copy and encodes the PNG while the next frame renders."""

    def __init__(self, output_dir, exp_name, stages=4):
        import queue
        import threading
        self.image_dir = os.path.join(output_dir, exp_name)
        print('The rendering is saved in ' + self.image_dir)
        if os.path.exists(self.image_dir):
            shutil.rmtree(self.image_dir)
        os.makedirs(self.image_dir, exist_ok=True)
        self.frame_idx = -1
        self._jobs = queue.Queue()
        self._queue_cls = queue.Queue
        self._free = {}                                         # image shape -> queue of free pinned stages of that shape
        self._stages, self._n_stages = {}, stages
        self._error = None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _work(self):
        from PIL import Image
        while True:
            job = self._jobs.get()
            if job is None:
                return
            image, event, name, stage = job
            try:
                if event is not None:
                    event.synchronize()
                Image.fromarray(image.numpy() if torch.is_tensor(image) else image).save(f'{self.image_dir}/{name}.png')
            except Exception as e:                              # surfaced by finalize()
                self._error = e
            if stage is not None:
                self._free[tuple(stage.shape)].put(stage)

    def _name(self, img_name):
        self.frame_idx += 1
        return f'{self.frame_idx:06d}' if img_name is None else img_name

    def append(self, image, img_name=None):
        name = self._name(img_name)
        self._jobs.put((np.ascontiguousarray(image), None, name, None))
        return self.frame_idx, name

    def append_device(self, image_u8, img_name=None):
        """image_u8: uint8 [H,W,3] on the GPU (assemble_uint8_device).  Returns at once."""
        name = self._name(img_name)
        key = tuple(image_u8.shape)
        pool = self._stages.setdefault(key, [])
        free = self._free.setdefault(key, self._queue_cls())    # one queue per shape: frames of mixed sizes never starve
        if len(pool) < self._n_stages:
            stage = torch.empty(key, dtype=torch.uint8).pin_memory()
            pool.append(stage)
        else:
            stage = free.get()                                  # blocks only when the encoder is `stages` frames behind
        stage.copy_(image_u8, non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(image_u8.device))
        self._jobs.put((stage, event, name, stage))
        return self.frame_idx, name

    def finalize(self):
        self._jobs.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error
