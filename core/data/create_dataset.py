"""create_dataloader(data_type): the reference's data entry point (core/data/create_dataset.py:
59-74).  The ZJU-MoCap / OcMotion pickles and the SMPL model are not redistributable, so this
build ships only the synthetic subject (occnerf_amd/synth.py): same per-frame dict, same shapes,
yielded with the leading batch dimension a torch DataLoader with batch_size=1 would add
(run.py strips it, run.py:85-86)."""
import numpy as np
import torch

from configs import cfg
from occnerf_amd import synth


class SyntheticFrames:
    """tpose: 1 frame, zero pose; freeview: cfg.render_frames orbit frames of one seeded pose;
    movement: cfg.render_frames frames of a seeded smooth pose walk from one camera; allview: the freeview pose from
    the 23 cameras of a ZJU-MoCap-like ring (allview.py:69); progress: up to 300 frames of the walk, the camera moving
    with them (create_dataset.py:40-42 `maxframes = 300` under evaluate)."""

    def __init__(self, data_type):
        self.data_type = data_type
        self.img_size = int(cfg.get('render_size', 512))
        self.avg_betas = np.zeros(10, dtype='float32')
        self.total_frames = {'tpose': 1, 'allview': 23, 'progress': min(300, int(cfg.render_frames))}.get(
            data_type, int(cfg.render_frames))
        self.dataset = self                     # run.py reads test_loader.dataset.avg_betas

    def __len__(self):
        return self.total_frames

    def _pose(self, idx):
        if self.data_type == 'tpose':
            return None
        if self.data_type in ('movement', 'progress'):
            return synth.movement_pose(idx, self.total_frames)
        return synth.seeded_pose(int(cfg.freeview.get('frame_idx', 0)) + 1)

    def __iter__(self):
        for idx in range(self.total_frames):
            frame = synth.make_frame(
                img_size=self.img_size, pose72=self._pose(idx),
                orbit_frame=idx if self.data_type in ('freeview', 'allview', 'progress') else 0,
                orbit_period=max(self.total_frames, 1), bgcolor=cfg.bgcolor,
                with_rays=not bool(cfg.get('device_rays', True)))
            batch = {}
            for k, v in frame.items():
                batch[k] = torch.as_tensor(np.asarray(v))[None] if not np.isscalar(v) else v
            batch['frame_name'] = [f'frame_{idx:06d}']
            yield batch


def create_dataloader(data_type='train', evaluate=False, **_):
    if cfg.get('dataset', 'synthetic') != 'synthetic' or \
            data_type not in ('tpose', 'freeview', 'movement', 'allview', 'progress'):
        raise NotImplementedError(
            f"dataset '{cfg.get('dataset')}' / type '{data_type}': only the synthetic tpose / freeview / movement / "
            'allview / progress frame generators ship with this build (datasets are out of scope, SURVEY.md 2)')
    return SyntheticFrames(data_type)
